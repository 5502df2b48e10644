"""The level-1 post-processor programs for blocks above 16 MiB (rb > 0 raw offset bits) and for E8E9 blocks have no
byte fixture in the reference tree (parity unpinned).  They are pinned by DECODE parity instead: the
REAL reference LZBuffer (ZSFX/libzpaq.cpp:6140-6552) produces the code stream, the program this engine's makeConfig +
compiler emit is put in front of it, and the REAL reference PostProcessor / ZPAQL machine (:2178-2233) must give the
input back.  The same programs are the ones the decode side recognises and runs natively (zpq_known_pcomp_bytes)."""
import ctypes as C

import numpy as np
import pytest

import datagen
import orc
from zpaqfranz_amd import engine

pytestmark = pytest.mark.ref


def known(rb, e8):
    L = engine.load()
    L.zpq_known_pcomp_bytes.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = (C.c_ubyte * 2048)(); n = C.c_size_t(0)
    assert L.zpq_known_pcomp_bytes(rb, int(e8), buf, 2048, C.byref(n)) == 0
    return bytes(buf[: n.value])


def exe_like(n, seed):
    a = bytearray(datagen.binary_like(n, seed))
    rng = np.random.default_rng(seed)
    for p in rng.integers(0, n - 8, n // 40):
        a[p] = 0xE8 if p & 2 else 0xE9
        a[p + 4] = 0 if p & 1 else 0xFF
    a[1000:1040] = b"\xe8" * 40                      # dense opcode run: chains of rewrites
    return bytes(a)


@pytest.mark.parametrize("arg0,e8", [(4, False), (4, True), (5, False), (5, True), (6, True), (7, False)])
def test_reference_vm_restores_the_input_under_our_program(arg0, e8):
    method = "x%d,%d,5,0,3,%d" % (arg0, 5 if e8 else 1, min(26, 19 + arg0 + (1 if arg0 <= 6 else 0)))
    src, args = engine.make_config(method)
    hdr, pc = engine.compile_config(src, args)
    assert pc == known(max(arg0 - 4, 0), e8)
    n = 300000
    data = exe_like(n, 5) if e8 else datagen.mixed(n, 6)
    lz = orc.ref_lzbuffer(data, args)                 # args[1] = 5: the reference applies e8e9() itself
    stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + lz
    assert orc.ref_postprocess(stream, hdr[4], hdr[5], n + 64) == data


def test_golden_program_is_variant_zero():
    import re, os
    src = open(os.path.join(orc.ROOT, "zpaqfranz_amd", "csrc", "block.hip")).read()
    m = re.search(r"zpq_pcomp_lz1\[302\] = \{(.*?)\};", src, re.S)
    gold = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", m.group(1)))
    assert len(gold) == 302 and known(0, False) == gold
    blocks = open(os.path.join(orc.GOLDEN, "sha256.zpaq"), "rb").read()
    assert gold in blocks                              # it is what the reference's own i blocks carry


# ---- levels 2 and 3 (what methods 3 and 4 put in front of their models): byte-aligned LZ77 and BWT ----------
@pytest.mark.parametrize("arg0,e8", [(0, False), (4, False), (4, True), (6, False)])
def test_reference_vm_undoes_byte_aligned_lz77_under_our_program(arg0, e8):
    """REAL LZBuffer at level 2 (suffix-array match finder, as "x<N>,2,12,0,7,<21+N>,1" asks) -> our level-2 program in
    front -> the REAL PostProcessor must give the input back.  The program without E8E9 has the 108 bytes libzpaq's own
    HCOMP prelude counts on (it skips 3 + 108 bytes of preamble before the first LZ77 code)."""
    method = "x%d,%d,12,0,7,%d,1c0,0,511i2" % (arg0, 6 if e8 else 2, 21 + arg0)
    src, args = engine.make_config(method)
    hdr, pc = engine.compile_config(src, args)
    assert len(pc) == (164 if e8 else 108)          # (the HCOMP prelude of level 2 skips 3 + len(pc) bytes: config.hip)
    n = 200000
    data = exe_like(n, 7) if e8 else datagen.mixed(n, 8)
    lz = orc.ref_lzbuffer(data, args)
    stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + lz
    assert orc.ref_postprocess(stream, hdr[4], hdr[5], n + 64) == data


@pytest.mark.parametrize("arg0,e8", [(0, False), (4, False), (4, True), (5, False), (5, True), (6, True)])
def test_reference_vm_inverts_the_bwt_under_our_program(arg0, e8):
    """REAL LZBuffer at level 3 (divbwt + index) -> our BWT program -> the REAL PostProcessor restores the input."""
    method = "x%d,%dci1" % (arg0, 7 if e8 else 3)
    src, args = engine.make_config(method)
    hdr, pc = engine.compile_config(src, args)
    n = 150000
    data = exe_like(n, 9) if e8 else datagen.text_like(n, 10)
    bw = orc.ref_lzbuffer(data, args)
    assert len(bw) == len(data) + 5
    stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + bw
    assert orc.ref_postprocess(stream, hdr[4], hdr[5], n + 64) == data
    for edge in (b"", b"a", b"abracadabra"):
        bw = orc.ref_lzbuffer(edge, args)
        stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + bw
        assert orc.ref_postprocess(stream, hdr[4], hdr[5], 64) == edge


@pytest.mark.parametrize("method", ["x0,6,12,0,7,21,1c0,0,511i2", "x0,7ci1", "x4,7ci1", "x5,7ci1", "x0,4c0", "x0,5,5,0,3,20", "x5,5,5,0,3,25"])
def test_e8_e9_among_the_last_bytes_is_left_alone_by_our_programs(method):
    """e8e9() (ZSFX/libzpaq.cpp:6117-6126) starts at n - 5: an E8 / E9 among the last four bytes has no transformed operand.  Until
    round 6 the E8E9 stage of the level-2, level-3 and E8E9-only programs tested byte i + 4 BEHIND the data there (zeros / stale
    bytes of M pass the test) and the reference's PostProcessor gave back other bytes than went in.  Every tail shape, every
    program with an E8E9 stage, under the REAL PostProcessor."""
    src, args = engine.make_config(method)
    hdr, pc = engine.compile_config(src, args)
    for seed in range(6):
        for tail in ([1], [2], [3], [4], [5], [2, 3, 4], [1, 2, 3, 4, 5, 6, 7, 8]):
            d = bytearray(datagen.binary_like(2500 + 7 * seed, 40 + seed))
            for t in tail:
                d[-t] = 0xE8 + (t & 1)
            d = bytes(d)
            body = orc.ref_lzbuffer(d, args) if (args[1] & 3) else orc.ref_e8e9(d)
            stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + body
            assert orc.ref_postprocess(stream, hdr[4], hdr[5], len(d) + 64) == d, (seed, tail)
    # a buffer shorter than five bytes, and an empty one
    for d in (b"", b"\xe8", b"\xe8\x00\x00\x00", b"\xe9\xe8\xe8\xe8\xe8"):
        body = orc.ref_lzbuffer(d, args) if (args[1] & 3) else orc.ref_e8e9(d)
        stream = bytes([1, len(pc) & 255, len(pc) >> 8]) + pc + body
        assert orc.ref_postprocess(stream, hdr[4], hdr[5], len(d) + 64) == d
