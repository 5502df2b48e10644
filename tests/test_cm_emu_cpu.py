"""The run-time specialised context-mixing coder on the CPU.  For a block header cm_jit.hip GENERATES HIP source (lane masks,
the dependent chain with constant lanes, HCOMP as straight-line code); on the GPU hiprtc compiles it.  Here the same text --
zpq_cm_spec_source_text(), with one line changed: the v_writelane inline assembly becomes C -- is compiled for the host
between tests/cpp/cm_emu_head.inc and cm_emu_tail.inc and run on the fibre emulator (simt_emu.h): a 16-wave workgroup, its
LDS tables, DPP sums, readlanes, the queue.  The coded bytes must be the REFERENCE Predictor's (oracle/_ref), and decode
back.  Covers every component type, the models of methods 3 / 4 / 5 and libzpaq's three built-in models."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cmconfigs
import datagen
import orc
from zpaqfranz_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = [pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found"),
              pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")]


def build(L, header, path):
    buf = C.create_string_buffer(1 << 20)
    n = C.c_size_t()
    assert L.zpq_cm_spec_source_text(header, len(header), buf, 1 << 20, C.byref(n)) == 0
    src = buf.raw[: n.value].decode()
    old, = [ln for ln in src.splitlines() if ln.startswith("#define ZWL(") and "v_writelane" in ln]
    src = src.replace(old, "#define ZWL(v, L, p) { const int zwl_ = __builtin_amdgcn_readfirstlane((int)(v)); if ((int)(threadIdx.x & 63) == (L)) p = (decltype(p))zwl_; }")
    cpp = os.path.join(path, "cm_emu.cpp")
    with open(cpp, "w") as f:
        f.write(open(os.path.join(ROOT, "tests", "cpp", "cm_emu_head.inc")).read() + src + open(os.path.join(ROOT, "tests", "cpp", "cm_emu_tail.inc")).read())
    so = os.path.join(path, "cm_emu.so")
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-Wno-unused-value", "-I" + os.path.join(ROOT, "tests", "cpp"), cpp, "-o", so])
    lib = C.CDLL(so)
    lib.cm_emu.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int,
                           C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    return lib


def code(lib, t, header, data, encode, cap, segments=None):
    """segments: the block's segments (their bytes, or their coded streams) instead of `data` -> the output of each"""
    if segments is not None:
        data = b"".join(segments)
    k = len(segments) if segments is not None else 0
    out = np.zeros(cap + 64, dtype=np.uint8)
    res = (C.c_uint32 * 2)()
    err = C.create_string_buffer(256)
    lens = (C.c_uint32 * max(1, k))(*[len(x) for x in (segments or [])]); ends = (C.c_uint32 * max(1, k))()
    rc = lib.cm_emu(header, t["sq"].ctypes.data, t["st"].ctypes.data, t["dt"].ctypes.data, t["dt2"].ctypes.data, t["ns"].ctypes.data, data + bytes(64), len(data),
                    out.ctypes.data, cap, 1 if encode else 0, res, err, 256, lens, k, ends)
    assert rc == 0, err.value.decode()
    assert res[1] == 0, "status %d" % res[1]
    if k > 1:
        cuts = [0] + [int(e) for e in ends]
        assert cuts[-1] == res[0], (cuts, res[0])
        return [bytes(out[cuts[i]:cuts[i + 1]]) for i in range(k)]
    return bytes(out[: res[0]])


def _header(L, name):
    if name in cmconfigs.ALL:
        return engine.compile_config(cmconfigs.ALL[name], [0] * 9)[0]
    if name.startswith("builtin"):
        buf = C.create_string_buffer(512)
        n = C.c_size_t(0)
        assert L.zpq_builtin_model(int(name[-1]), buf, 512, C.byref(n)) == 0
        return buf.raw[: n.value]
    src, args = engine.make_config(engine.expand_method({"m4": "44", "m5": "54", "m3bwt": "x4,3ci1"}[name], b"x" * 1000))
    return engine.compile_config(src, args)[0]


MODELS = [("order1_cm", 600), ("mid", 300), ("alltypes", 300), ("m4", 300), ("m5", 160), ("builtin1", 600), ("builtin3", 160), ("m3bwt", 600)]


def _one_model(job):
    """worker process: generate, compile, code four inputs, decode them back; returns a text on any difference"""
    name, nbytes, path = job
    L = engine.load()
    L.zpq_cm_spec_source_text.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_cm_tables.argtypes = [C.c_void_p] * 5
    L.zpq_builtin_model.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    t = dict(sq=np.zeros(4096, dtype=np.uint16), st=np.zeros(32768, dtype=np.int16), dt=np.zeros(1024, dtype=np.int32), dt2=np.zeros(256, dtype=np.int32),
             ns=np.zeros(1024, dtype=np.uint8))
    L.zpq_cm_tables(t["sq"].ctypes.data, t["st"].ctypes.data, t["dt"].ctypes.data, t["dt2"].ctypes.data, t["ns"].ctypes.data)
    try:
        h = _header(L, name)
        os.makedirs(path, exist_ok=True)
        lib = build(L, h, path)
        for x in (b"\0" + datagen.text_like(nbytes, 1), b"\0" + datagen.binary_like(nbytes // 2, 2), b"\0", b""):
            got = code(lib, t, h, x, True, len(x) * 2 + 4096)
            if got != orc.ref_cm_encode(h, x):
                return "%s: %d bytes code differently from the reference Predictor" % (name, len(x))
            if code(lib, t, h, got, False, len(x) + 64) != x:
                return "%s: %d bytes do not decode back" % (name, len(x))
        # several segments in one block: the model carries on from one to the next (the real Predictor / Decoder, kept
        # across segments, are the measure), an empty segment in the middle and at the end
        segs = [b"\0" + datagen.text_like(nbytes // 2, 3), datagen.text_like(nbytes // 3, 4), b"", datagen.binary_like(nbytes // 4, 5), b""]
        want = orc.ref_cm_encode_segments(h, segs)
        got = code(lib, t, h, None, True, sum(map(len, segs)) * 2 + 4096, segments=segs)
        if got != want:
            return "%s: a block of %d segments codes differently from the reference Predictor" % (name, len(segs))
        if want[1] == orc.ref_cm_encode(h, segs[1]):
            return "%s: the second segment does not depend on the first (test input too weak)" % name
        if code(lib, t, h, None, False, sum(map(len, segs)) + 64, segments=want) != segs:
            return "%s: a block of %d segments does not decode back" % (name, len(segs))
    except Exception as ex:          # noqa: BLE001 -- reported by the parent
        return "%s: %r" % (name, ex)
    return None


def test_generated_kernels_on_the_emulator_code_like_the_reference_predictor(tmp_path):
    """one worker process per model (generate, compile, emulate), all at the same time"""
    import multiprocessing as mp
    jobs = [(name, n, str(tmp_path / name)) for name, n in MODELS]
    with mp.get_context("fork").Pool(min(len(jobs), os.cpu_count() or 2)) as pool:
        bad = [r for r in pool.map(_one_model, jobs) if r]
    assert not bad, bad
