"""The run-time specialisation of the context-mixing coder, as far as it goes without a GPU: for a block header the
generator (cm_jit.hip) emits HIP source -- lane masks, the dependent-component chain with constant lane numbers, the
HCOMP program as straight-line code -- and hiprtc cross-compiles it for gfx950.  The compiled kernels are exercised by
tests/test_gpu_cm_spec.py; here: every model of the test set and of methods 4 / 5 generates and compiles, the HCOMP
translation covers every reachable instruction, and hostile programs translate into bounded code."""
import ctypes as C
import os

import pytest

import cmconfigs
from zpaqfranz_amd import engine


@pytest.fixture(scope="module")
def L():
    lib = engine.load()
    lib.zpq_cm_precompile.argtypes = [C.c_char_p, C.c_uint32]
    lib.zpq_cm_spec_source_text.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return lib


def source(L, header):
    buf = C.create_string_buffer(1 << 20); n = C.c_size_t()
    assert L.zpq_cm_spec_source_text(header, len(header), buf, 1 << 20, C.byref(n)) == 0
    return buf.raw[: n.value].decode()


def method_header(m, data=b"x" * 1000):
    src, args = engine.make_config(engine.expand_method(m, data))
    return engine.compile_config(src, args)[0]


def raw_header(hh, hm, comps, hcomp):
    body = bytes([hh, hm, 0, 0, len(comps)]) + b"".join(bytes(c) for c in comps) + b"\0" + bytes(hcomp) + b"\0"
    return bytes([len(body) & 255, len(body) >> 8]) + body


@pytest.mark.parametrize("name", list(cmconfigs.ALL) + ["m4", "m5", "m3bwt", "m3lz"])
def test_generated_kernels_compile_for_gfx950(L, name, tmp_path, monkeypatch):
    monkeypatch.setenv("ZPQ_JIT_CACHE", str(tmp_path))
    if name in cmconfigs.ALL:
        h = engine.compile_config(cmconfigs.ALL[name], [0] * 9)[0]
    else:
        h = method_header({"m4": "44", "m5": "54", "m3bwt": "x4,3ci1", "m3lz": "x4,2,12,0,7,25,1c0,0,511i2"}[name])
    src = source(L, h)
    assert "#define ZN %d\n" % h[6] in src and "z_hcomp" in src
    assert L.zpq_cm_precompile(h, len(h)) == 0
    assert any(f.endswith(".hsaco") for f in os.listdir(tmp_path))       # the code object landed in the cache directory


def test_chain_of_isse_fed_by_neighbours_is_grouped(L):
    """mid: icm, five ISSEs each fed by its left neighbour, match, mix -> one group of depth 5, then the mixer."""
    src = source(L, engine.compile_config(cmconfigs.MID, [0] * 9)[0])
    chain = [ln for ln in src.splitlines() if ln.startswith("#define Z_CHAIN")][0]
    assert "Z_ISSE_SYS(0x3eull,5)" in chain and chain.index("Z_ISSE_SYS") < chain.index("Z_MIX(")
    assert "#define Z_ISSE_FAR_INPUTS \n" in src                          # no ISSE with a far input


def test_hcomp_translation_follows_jumps_into_operands_and_bounds_loops(L):
    # *d=a ; jmp +1 ; a= 56 : the jump lands on the operand byte 56 = halt
    src = source(L, raw_header(2, 4, [(2, 16, 255)], [112, 63, 1, 71, 56]))
    body = src[src.index("void z_hcomp"):]
    assert "L4: goto Lend;" in body and "L3:" not in body                 # pc 3 (a= 56) is never reached, pc 4 is the halt
    # jmp to itself: a counted backward jump
    src = source(L, raw_header(2, 4, [(2, 16, 255)], [63, 254]))
    # (2^24 backward jumps free per byte + the block's credit: gen_zpaql)
    assert "if (++guard > zlim) goto Llim; goto L0;" in src and "const u32 zlim = (u32)(ZGUARD) + z.credit;" in src
    # running off the end / invalid opcode -> error exit
    src = source(L, raw_header(2, 4, [(2, 16, 255)], [1, 5]))
    assert "L1: goto Lerr;" in src


def test_more_than_64_components_are_left_to_the_generic_kernel(L):
    comps = [(2, 8, 255)] * 65
    h = raw_header(2, 4, comps, [56])
    buf = C.create_string_buffer(1 << 16); n = C.c_size_t()
    assert L.zpq_cm_spec_source_text(h, len(h), buf, 1 << 16, C.byref(n)) == -5
