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
