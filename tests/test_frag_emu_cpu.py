"""The fragmenter kernels (zpaqfranz_amd/csrc/fragment.hip) on the CPU: tests/cpp/frag_emu.cpp compiles their DEVICE source for
the host and runs them on the fibre emulator (tests/cpp/simt_emu.h) exactly as fragment_run() launches them -- speculative
lanes with their LDS o1[] tables, the resume launch for parked crossing walks, the per-file stitch with the exact wave
evaluator (LDS lane masks, DPP affine-map scan) and the emit kernel, with and without twins.  Cuts must be the oracle's and,
on the reference's own d block, the 388 golden fragment records."""
import ctypes as C
import lzma
import os
import struct
import subprocess

import numpy as np
import pytest

import datagen
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")


@pytest.fixture(scope="module")
def frag(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("frag") / "frag_emu.so")
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "frag_emu.cpp"), "-o", so])
    L = C.CDLL(so)
    L.frag_emu.restype = C.c_long
    L.frag_emu.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint32]

    def run(files, frag=6, minf=4096, maxf=520192, seg=65536, waves=2, budget=262144, rep=None):
        off = [0]
        for f in files:
            off.append(off[-1] + len(f))
        cap = sum(len(f) // minf + 1 for f in files) + 4
        fo, fl, ff = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        err = C.create_string_buffer(256)
        reparr = np.array(rep, dtype=np.uint32) if rep is not None else None
        r = L.frag_emu(b"".join(files) + bytes(64), (C.c_uint64 * len(off))(*off), len(files), frag, minf, maxf, seg, waves, budget,
                       reparr.ctypes.data if rep is not None else None, fo.ctypes.data, fl.ctypes.data, ff.ctypes.data, cap, err, 256)
        assert r >= 0, (r, err.value.decode())
        return [(int(ff[i]), int(fo[i]) - off[int(ff[i])], int(fl[i])) for i in range(r)]
    return run


def oracle(files, frag=6, minf=4096, maxf=520192):
    out = []
    for fi, f in enumerate(files):
        o = 0
        for ln in orc.chunk(f, frag, minf, maxf):
            out.append((fi, o, ln))
            o += ln
    return out


FILES = [datagen.mixed(300000, 1), datagen.text_like(150000, 2), b"", datagen.binary_like(80000, 3), bytes(70000), b"ab" * 40000, b"a", datagen.random_bytes(4096, 4),
         datagen.random_bytes(4097, 5), b"x" * 4095, datagen.text_like(65536, 6), datagen.text_like(65537, 7)]


@pytest.mark.parametrize("kw", [dict(), dict(seg=16384, waves=1), dict(seg=86016, waves=3), dict(budget=4096), dict(budget=1 << 40), dict(seg=32768, budget=1000, waves=2)],
                         ids=["default", "seg16k-1wave", "seg84k", "park-nearly-all", "park-none", "small-budget"])
def test_emulated_fragmenter_gives_the_oracles_cuts(frag, kw):
    assert frag(FILES, **kw) == oracle(FILES)


@pytest.mark.parametrize("fragment,minf,maxf", [(0, 64, 8128), (3, 512, 65024), (8, 16384, 2080768)])
def test_emulated_fragmenter_other_settings(frag, fragment, minf, maxf):
    files = [datagen.mixed(400000, 21), datagen.text_like(60000, 22), datagen.binary_like(150000, 23)]
    assert frag(files, frag=fragment, minf=minf, maxf=maxf, seg=max(65536, 4 * minf)) == oracle(files, fragment, minf, maxf)


def test_emulated_fragmenter_predictable_data_takes_the_wave_evaluator(frag):
    """Constant and periodic stretches: speculative and true chains never fall in step there, the exact wave evaluator
    (steady-state fast path and the general path at the edges of the runs) does the work."""
    rng = np.random.default_rng(31)
    parts = []
    for i in range(10):
        n = int(rng.integers(20000, 90000))
        parts.append([bytes([int(rng.integers(0, 256))]) * n, (bytes(rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8)) * n)[:n],
                      datagen.text_like(n // 8, 100 + i), (b"ab" * n)[:n], datagen.random_bytes(n // 16, 200 + i)][i % 5])
    big = b"".join(parts)
    files = [big, bytes(600000), big[12345:200000]]
    assert frag(files) == oracle(files)
    assert frag(files, seg=16384, waves=1, budget=8192) == oracle(files)


def test_emulated_fragmenter_twins_take_their_representatives_cuts(frag):
    a, b = datagen.mixed(300000, 41), datagen.text_like(90000, 42)
    files = [a, b, a, datagen.binary_like(50000, 43), a, b]
    assert frag(files, rep=[0, 1, 0, 3, 0, 1]) == oracle(files)


def test_emulated_fragmenter_reproduces_the_golden_fragment_records(frag):
    """The reference's own d block (AUTOTEST/sha256.zpaq): 256 files of 37000 bytes, 388 fragments -- sizes and SHA-1s of the
    golden h block."""
    dplain = lzma.decompress(open(os.path.join(orc.GOLDEN, "dblock_plain.xz"), "rb").read())
    h = open(os.path.join(orc.GOLDEN, "hblock_plain.bin"), "rb").read()
    want = [(h[4 + 24 * i: 24 + 24 * i], struct.unpack("<I", h[24 + 24 * i: 28 + 24 * i])[0]) for i in range(388)]
    files = [dplain[i * 37000:(i + 1) * 37000] for i in range(256)]
    got = frag(files, waves=4)
    assert [ln for _, _, ln in got] == [u for _, u in want]
    assert [orc.sha1(files[f][o:o + ln]) for f, o, ln in got] == [s for s, _ in want]
