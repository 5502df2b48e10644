"""The candidate-table formulation of LZBuffer's hash-table search (zpaqfranz_amd/csrc/lz77_enc.hip, experimental), checked
on the CPU: every position is inserted whatever the parse decides (ZSFX/libzpaq.cpp:6432-6447), so the bucket+1 table
words a search reads at position q can be computed for every q in advance and the parse can run from them alone.  The
oracle's two restatements must agree, and -- where oracle/_ref is available -- equal the real LZBuffer's stream."""
import numpy as np
import pytest

import datagen
import orc

INPUTS = {
    "text": datagen.text_like(120000, 1),
    "binary": datagen.binary_like(90000, 2),
    "mixed": datagen.mixed(200000, 3),
    "runs": bytes(30000) + b"ab" * 20000 + datagen.random_bytes(3000, 4),
    "tiny": b"abcabcabcabc", "one": b"x", "empty": b"", "nine": b"123456789",
}
ARGS = [[4, 1, 5, 0, 3, 24], [0, 1, 4, 0, 1, 15], [4, 1, 4, 0, 2, 16], [0, 1, 6, 0, 3, 20], [4, 1, 5, 0, 0, 22], [5, 1, 4, 0, 2, 18]]


@pytest.mark.parametrize("args", ARGS, ids=lambda a: ",".join(map(str, a)))
def test_parse_from_candidate_tables_equals_the_table_parse(args):
    for name, b in INPUTS.items():
        cand = orc.lz77_cand(b, args)
        assert len(cand) == len(b) << args[4]
        assert orc.lz77_encode_from_cand(b, args, cand) == orc.lz77_encode(b, args), name


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")
def test_parse_from_candidate_tables_equals_the_real_lzbuffer():
    b = INPUTS["mixed"]
    args = [4, 1, 5, 0, 3, 24]
    assert orc.lz77_encode_from_cand(b, args, orc.lz77_cand(b, args)) == orc.ref_lzbuffer(b, args)


def test_candidate_table_is_what_a_group_sweep_gives():
    """The GPU builds the table by sorting positions by (hash group, position) and sweeping every group once (a lane keeps
    the group's bucket+1 words, writes them out in probe order at each position, then applies that position's insert).
    The same in numpy / Python on a small input must give the sequential table."""
    b = datagen.mixed(30000, 8)
    args = [4, 1, 5, 0, 3, 18]
    want = orc.lz77_cand(b, args)
    n, mm, lb, htbits, cb = len(b), args[2], args[4], args[5], 12 - args[0]
    B, mask, msk = (1 << lb) - 1, (1 << htbits) - 1, (1 << cb) - 1
    upd = n - (mm + 4) if n > mm + 4 else 0
    shift1 = (htbits - 1) // mm + 1
    F = (5 << shift1) & 0xFFFFFFFF
    a = np.frombuffer(b, dtype=np.uint8)
    h = np.zeros(n, dtype=np.uint64)
    for q in range(n):                       # h1 as LZBuffer holds it at q: over in[U-mm+mm .. U-1+mm], U = min(q, upd)
        U = min(q, upd)
        x = 0
        for t in range(max(0, U - mm), U):
            x = (x * F + (int(a[t + mm]) + 1) * 123456791) & 0xFFFFFFFF
        h[q] = x & mask
    order = np.lexsort((np.arange(n), h >> np.uint64(lb)))       # by (group, position)
    got = np.zeros(n << lb, dtype=np.uint32)
    v, g_prev = [0] * (B + 1), None
    for q in order.tolist():
        g = int(h[q]) >> lb
        if g != g_prev:
            v, g_prev = [0] * (B + 1), g
        hb = int(h[q]) & B
        for k in range(B + 1):
            got[(q << lb) + k] = v[hb ^ k]
        if q < upd:
            v[hb ^ (((q * 1234547) & 0xFFFFFFFF) >> 19 & B)] = ((q << cb) & 0xFFFFFFFF) | (int(a[q + 3]) & msk)
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# The DEVICE source of the candidate-table kernels (zpaqfranz_amd/csrc/lz77_cand.inc) compiled for the host under a serial
# SIMT shim (tests/cpp/cand_host.cpp): the same code the GPU runs, one thread after the other, must give the oracle's table.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cand_host(tmp_path_factory):
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("cand") / "cand_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I" + os.path.join(root, "zpaqfranz_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "cand_host.cpp"), "-o", so])
    L = C.CDLL(so)
    L.cand_host.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int32), C.c_void_p]

    def run(blocks, args):
        off, pos = [], 0
        for b in blocks:
            off.append(pos); pos += len(b) + 3            # odd spacing: blocks at every alignment
        buf = bytearray(pos + 16)
        for o, b in zip(off, blocks):
            buf[o:o + len(b)] = b
        words = sum(len(b) << args[4] for b in blocks)
        cand = np.zeros(max(1, words), dtype=np.uint32)
        rc = L.cand_host(bytes(buf), (C.c_uint64 * len(blocks))(*off), (C.c_uint32 * len(blocks))(*[len(b) for b in blocks]), len(blocks),
                         (C.c_int32 * 9)(*(list(args) + [0] * 9)[:9]), cand.ctypes.data_as(C.c_void_p))
        assert rc == 0
        out, w = [], 0
        for b in blocks:
            out.append(cand[w:w + (len(b) << args[4])]); w += len(b) << args[4]
        return out
    return run


@pytest.mark.parametrize("args", ARGS, ids=lambda a: ",".join(map(str, a)))
def test_device_source_of_the_candidate_kernels_on_the_host(cand_host, args):
    names = list(INPUTS)
    got = cand_host([INPUTS[k] for k in names], args)           # all inputs as ONE batch of blocks
    for k, g in zip(names, got):
        want = orc.lz77_cand(INPUTS[k], args)
        bad = np.nonzero(g != want)[0]
        assert bad.size == 0, (k, int(bad[0]) >> args[4], int(bad[0]) & ((1 << args[4]) - 1), int(g[bad[0]]), int(want[bad[0]]), bad.size)
    one, = cand_host([INPUTS["mixed"]], args)                    # and a block on its own
    assert np.array_equal(one, orc.lz77_cand(INPUTS["mixed"], args))
