"""tools/proto/zpaql_to_c.py: a ZPAQL program translated once into straight-line C must behave
like the reference's interpreter.  The generated C is compiled with gcc and driven like PostProcessor::write drives a
PCOMP program (once per byte, then once with 2^32-1); the OUT bytes are compared with the REAL reference VM
(oracle/_ref, ref_postprocess)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "proto"))
pytestmark = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")

HARNESS = r'''
#include <stdlib.h>
#include <string.h>
static unsigned char* g_out; static long g_n, g_cap;
static void put(void* a, int c) { (void)a; if (g_n < g_cap) g_out[g_n] = (unsigned char)c; ++g_n; }
long drive(const unsigned char* in, long n, int ph, int pm, unsigned char* out, long cap) {
  struct zvm z; memset(&z, 0, sizeof z);
  z.h = calloc((size_t)1 << ph, 4); z.hmask = (1u << ph) - 1; z.m = calloc((size_t)1 << pm, 1); z.mmask = (1u << pm) - 1;
  z.out = put; g_out = out; g_n = 0; g_cap = cap;
  for (long i = 0; i < n && !z.err; ++i) zpaql_run(in[i], &z);
  if (!z.err) zpaql_run(0xffffffffu, &z);
  free(z.h); free(z.m);
  return z.err ? -1 : g_n;
}
'''

PROGRAMS = {
    # the LZ77 level-1 post-processor of the fixtures (302 bytes), via the product's own config path
    "lazy2": None,
    # every opcode group, both jump forms, H and M traffic, division by zero, the byte-only swap
    "allops": "comp 0 0 3 8 0 hcomp halt pcomp x ; "
              "a> 255 if halt endif b=a c=a d=a *b=a *c=a a+= 3 *d=a a=*b a<>a? ".replace("a<>a? ", "") +
              "b<>a c<>a d<>a *b<>a *c<>a *d<>a a! b! c! d! *b! *c! *d! a++ b-- c++ d-- *b++ *c-- *d++ a=0 hash hashd r=a 7 a=r 7 b=r 7 "
              "a+=b a-=c a*=d a/= 0 a%= 0 a/=b a%=c a&=d a&~b a|= 5 a^=*c a<<=b a>>= 3 a==*d jt 2 a++ a<c jf 1 out a>*b "
              "ifl a+= 1 out elsel a-= 1 out endif do a++ out a< 9 while a=*d out *c=0 *d=0 out halt end",
}


def build(tmp_path, name, code):
    import zpaql_to_c
    src = zpaql_to_c.emit_c(code) + HARNESS
    c = tmp_path / (name + ".c")
    so = tmp_path / (name + ".so")
    c.write_text(src)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so), str(c)])
    lib = C.CDLL(str(so))
    lib.drive.restype = C.c_long
    lib.drive.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, C.c_char_p, C.c_long]
    return lib


@pytest.mark.parametrize("name", list(PROGRAMS))
def test_translated_program_equals_reference_vm(tmp_path, name):
    from zpaqfranz_amd import build as zbuild, engine
    zbuild.build(verbose=False)
    if name == "lazy2":
        src, args = engine.make_config("x4,1,5,0,3,24")
        ph, pm = 0, 24
    else:
        src, args = PROGRAMS[name], [0] * 9
        ph, pm = 3, 8
    _, pcomp = engine.compile_config(src, args)
    lib = build(tmp_path, name, pcomp)
    rng = np.random.default_rng(5)
    if name == "lazy2":
        import datagen
        plain = datagen.text_like(60000, 7) + bytes(3000) + datagen.random_bytes(2000, 8)
        data = orc.lz77_encode(plain, [4, 1, 5, 0, 3, 24])
    else:
        data = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    cap = 1 << 20
    out = C.create_string_buffer(cap)
    n = lib.drive(data, len(data), ph, pm, out, cap)
    stream = b"\x01" + bytes([len(pcomp) & 255, len(pcomp) >> 8]) + pcomp + data
    want = orc.ref_postprocess(stream, ph, pm, cap)
    assert n == len(want) and out.raw[:n] == want
    if name == "lazy2":
        assert want == plain
