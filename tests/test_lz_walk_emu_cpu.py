"""The LZ77 hash-table parse (zpaqfranz_amd/csrc/lz77_enc.hip, lz77_waves.inc) run on the CPU: tests/cpp/walk_emu.cpp compiles
THE DEVICE SOURCE for the host and runs it as emulated waves (64 lanes as fibres in lockstep, the wave-level operations as
rendezvous; the three waves of a parse workgroup -- producer | evaluator | chain -- as 192 fibres whose spin loops yield to
each other).  Tokens and code streams must be the oracle's.  Needs the ROCm clang++ (address spaces, ext vectors) as a
host compiler."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import datagen
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")


_SO = [None]


def _lib(tmp_path_factory):
    """walk_emu.so, built once per test process"""
    if _SO[0] is None:
        so = str(tmp_path_factory.mktemp("walk_emu") / "walk_emu.so")
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused"] + os.environ.get("EMU_FLAGS", "").split() + ["-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"),
                               os.path.join(ROOT, "tests", "cpp", "walk_emu.cpp"), "-o", so])
        _SO[0] = C.CDLL(so)
    return _SO[0]


def _inputs():
    rng = np.random.default_rng(9)
    unit = rng.integers(0, 256, size=9000, dtype=np.uint8).tobytes()
    return {
        "text": datagen.text_like(12000, 1),
        "binary": datagen.binary_like(8000, 2),
        "mixed": datagen.mixed(10000, 3),
        "runs": bytes(4000) + b"ab" * 2500 + datagen.random_bytes(600, 4),            # very long matches: whole-wave compares
        "long": unit + unit + unit[:100] + datagen.random_bytes(6000, 5) + unit,       # capped candidates, a literal run past 4096
        # literal runs of 250..530 bytes between matches: the emitting wave's byte ring filled across chunk borders, the 512-byte limit of its LDS path
        "gaps": b"".join(unit[:700] + datagen.random_bytes(250 + 40 * k, 20 + k) for k in range(8)),
        "tiny": b"abcabcabcabc", "one": b"x", "empty": b"", "window": datagen.text_like(64, 7), "window+1": datagen.text_like(65, 8),
    }


ARGS = [[4, 1, 5, 0, 3, 16], [0, 1, 4, 0, 1, 15], [4, 1, 4, 0, 2, 16], [5, 1, 6, 0, 0, 15]]


def _pool_map(fn, jobs):
    """the cases of a test in forked worker processes (each has its own emulator state; the harness is already loaded)"""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(len(jobs), os.cpu_count() or 2)) as pool:
        return [r for r in pool.map(fn, jobs) if r]


def _walk_case(args):
    L = _SO[0]
    L.walk_emu.restype = C.c_long
    L.walk_emu.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    for name, b in _inputs().items():
        n = len(b)
        words = 1 << args[5]
        raw = np.zeros(words + 16, dtype=np.uint32)
        off = (-(raw.ctypes.data // 4)) % 4                      # 16-byte aligned view
        tab = raw[off:off + words]
        cap = n // 4 + 16
        tok = np.zeros(3 * cap, dtype=np.uint32)
        err = C.create_string_buffer(256)
        r = L.walk_emu(b + bytes(64), n, (C.c_int32 * 9)(*(list(args) + [0] * 9)[:9]), tab.ctypes.data, tok.ctypes.data, cap, err, 256)
        if r < 0:
            return (args, name, err.value.decode())
        got = [(int(tok[i]), int(tok[cap + i]), int(tok[2 * cap + i])) for i in range(r)]
        want = orc.lz77_encode(b, args, trace=True)[1]
        if got != want:
            return (args, name, len(got), len(want), next(((i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w), None))
    return None


def test_the_walk_on_an_emulated_wave_gives_the_oracles_tokens(tmp_path_factory):
    """lz_walk, the one-wave walk (what the seam and stitch kernels call)"""
    _lib(tmp_path_factory)
    bad = _pool_map(_walk_case, ARGS)
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------------------------
# The whole segment speculation of a block, kernel by kernel as encode_batch() launches them: table states (the copy and
# scatter kernels), lz77_spec3_kernel -- a workgroup of three waves per segment: producer | evaluator | chain with their
# rings in LDS (lz77_waves.inc) --, lz77_seam_kernel per segment, lz77_stitch_kernel, lz77_move_tokens_kernel.  Segments of a
# few KiB put seams, swallowed segments and re-walks into small inputs.
# ---------------------------------------------------------------------------------------------------------------------
SPEC_CASES = [([4, 1, 5, 0, 3, 15], 4096), ([0, 1, 4, 0, 1, 14], 8192), ([4, 1, 6, 0, 2, 15], 16384), ([5, 1, 5, 0, 0, 14], 4096),
              ([4, 2, 5, 0, 3, 15], 4096), ([0, 2, 4, 0, 1, 14], 16384)]          # level 2: the same parse with the byte codes' take rule


def _spec_case(job):
    args, seg = job
    L = _SO[0]
    L.spec_emu.restype = C.c_long
    L.spec_emu.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    inputs = _inputs()
    inputs["seams"] = (datagen.text_like(5000, 11) * 3)[:14000] + bytes(9000) + datagen.text_like(5000, 12)   # matches across several segment edges, a swallowed segment
    if (args[1] & 3) == 2:          # short matches at distances beyond 2^16: a far match must be longer to be taken
        rng = np.random.default_rng(21)
        base = rng.integers(0, 256, size=66000, dtype=np.uint8).tobytes()
        far = bytearray(base)
        for i in range(150):
            p = int(rng.integers(0, 400))
            far += base[p:p + args[2] + (i % 3)] + bytes(rng.integers(0, 256, size=3, dtype=np.uint8))
        inputs = {"far": bytes(far), "text": inputs["text"], "runs": inputs["runs"], "tiny": inputs["tiny"], "empty": b""}
    for name, b in inputs.items():
        n = len(b)
        cap = n // 4 + 16
        tok = np.zeros(3 * cap, dtype=np.uint32)
        err = C.create_string_buffer(256)
        r = L.spec_emu(b + bytes(64), n, (C.c_int32 * 9)(*(list(args) + [0] * 9)[:9]), seg, tok.ctypes.data, cap, err, 256)
        if r < 0:
            return (args, seg, name, r, err.value.decode())
        got = [(int(tok[i]), int(tok[cap + i]), int(tok[2 * cap + i])) for i in range(r)]
        want = orc.lz77_encode(b, args, trace=True)[1]
        if got != want:
            return (args, seg, name, len(got), len(want), next(((i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w), None))
    return None


def test_segment_speculation_on_emulated_waves_gives_the_oracles_tokens(tmp_path_factory):
    _lib(tmp_path_factory)
    bad = _pool_map(_spec_case, SPEC_CASES)
    assert not bad, bad


def _direct_case(args):
    L = _SO[0]
    L.direct_emu.restype = C.c_long
    L.direct_emu.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    for name, b in _inputs().items():
        n = len(b)
        words = 1 << args[5]
        raw = np.zeros(words + 16, dtype=np.uint32)
        off = (-(raw.ctypes.data // 4)) % 4
        tab = raw[off:off + words]
        cap = n + n // 8 + 1024
        out = np.zeros(cap, dtype=np.uint8)
        err = C.create_string_buffer(256)
        r = L.direct_emu(b + bytes(64), n, (C.c_int32 * 9)(*(list(args) + [0] * 9)[:9]), tab.ctypes.data, out.ctypes.data, cap, err, 256)
        if r < 0:
            return (args, name, r, err.value.decode())
        if bytes(out[:r]) != orc.lz77_encode(b, args):
            return (args, name, "stream differs")
    return None


def test_direct_kernel_on_emulated_waves_writes_the_oracles_stream(tmp_path_factory):
    """lz77_direct4_kernel: one workgroup of four waves parses a block and writes the code stream (the chain wave's matches
    go through a token ring to the emitting wave: literal runs spread over the lanes from its byte ring, match codes by put):
    byte for byte the oracle's stream, raw offset bits (args[0] > 4) included."""
    _lib(tmp_path_factory)
    bad = _pool_map(_direct_case, [[4, 1, 5, 0, 3, 15], [5, 1, 4, 0, 2, 15], [6, 1, 6, 0, 1, 14]])
    assert not bad, bad
