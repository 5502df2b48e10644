"""Rows a5/a6 on the CPU: the ZPAQL compiler against the REAL reference Compiler (oracle/_ref) and the
makeConfig/compressBlock restatement against the headers the reference's fixtures carry.

libzpaq's makeConfig() and compressBlock() are not in the snapshot (ZSFX/libzpaq.cpp ends after
LZBuffer), so the method strings and config sources are pinned on fixtures: level 1 on
AUTOTEST/sha256.zpaq (every block), level 5 on ZSFX/zsfx.zpaq and ZSFX/zsfx32.zpaq."""
import lzma
import os

import pytest

import cmconfigs
import orc

G = orc.GOLDEN
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")


@pytest.fixture(scope="module")
def cfg():
    from zpaqfranz_amd import build, engine
    build.build(verbose=False)
    return engine


def block_header(archive):
    k = archive.index(b"zPQ")
    hsize = archive[k + 5] | archive[k + 6] << 8
    return archive[k + 3], archive[k + 5:k + 7 + hsize], k + 7 + hsize


def ref_compiled(src, args):
    """The reference's pz.write(pp=true) puts a 2-byte length in front of the post-processor bytecode."""
    h, p = orc.ref_compile(src, args)
    if p:
        assert p[0] | p[1] << 8 == len(p) - 2
        p = p[2:]
    return h, p


LEVEL5 = "x0,0w1i1c256ci1,1,1,1,1,1,2ac0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0"


@pytest.mark.parametrize("name", ["zsfx", "zsfx32"])
def test_level5_fixture_header_reproduced(cfg, name):
    """method "5" on the fixture's own plaintext -> the x method, the config and the exact 255 header bytes."""
    arc = open(os.path.join(G, name + ".zpaq"), "rb").read()
    plain = lzma.decompress(open(os.path.join(G, name + "_plain.xz"), "rb").read())
    level, want, _ = block_header(arc)
    assert cfg.expand_method("5", plain) == LEVEL5
    src, args = cfg.make_config(LEVEL5)
    header, pcomp = cfg.compile_config(src, args)
    assert header == want and pcomp == b"" and level == 1


def test_journaling_fixture_headers_reproduced(cfg):
    """AUTOTEST/sha256.zpaq: c and h blocks are stored (method 0), the i blocks are method 1 (LZ77 + the 302-byte
    lazy2 post-processor) and the 9.4 MB d block is method 5 (23 components, 2^24 contexts)."""
    import json
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    plain = {"d": lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())}
    for b in json.load(open(os.path.join(G, "blocks.json"))):
        kind = b["filename"][17]
        blk = arc[b["offset"]:b["offset"] + b["size"]]
        level, want, p = block_header(blk)
        data = plain.get(kind, bytes(b["usize"]))
        assert len(data) == b["usize"]
        x = cfg.expand_method({"c": "0", "h": "0", "i": "1", "d": "5"}[kind], data)
        src, args = cfg.make_config(x)
        header, pcomp = cfg.compile_config(src, args)
        assert header == want, (b["filename"], x)
        assert level == (1 if kind == "d" else 2)
        if kind == "i":
            assert x == "x0,1,5,0,3,20"
            # segment header, then the stored stream opens with 1, len lo, len hi, pcomp
            q = blk.index(b"\0", blk.index(b"\0", p + 1) + 1) + 2        # after filename, comment, reserved 0
            body = blk[q + 4:]                                              # skip the first sub-block length
            assert body[0] == 1 and body[1] | body[2] << 8 == len(pcomp) == 302
            assert body[3:3 + 302] == pcomp
        if kind == "d":
            assert x == LEVEL5.replace("x0,", "x4,") and pcomp == b""


def test_method_expansion_table(cfg):
    n16 = bytes((1 << 24) - 4096)
    assert cfg.expand_method("14", n16) == "x4,1,5,0,3,24"
    assert cfg.expand_method("1", bytes(1000)) == "x0,1,5,0,3,20"
    assert cfg.expand_method("14,20,0", n16) == "x4,1,4,0,2,16"           # type 80
    assert cfg.expand_method("14,5,0", n16) == "x4,0"                      # type 20: store
    assert cfg.expand_method("14,255,0", n16) == "x4,1,6,0,3,24"          # type 1020
    assert cfg.expand_method("14,128,2", n16) == "x4,5,5,0,3,24"          # exe hint: E8E9 variant
    assert cfg.expand_method("0", bytes(5)) == "00,0"
    assert cfg.expand_method("4", bytes(1 << 20)) == "x1,0ci1,1,1,1,2am"
    assert cfg.expand_method("44,128,1", n16) == "x4,0ci1,1,1,1,2awm"
    assert cfg.expand_method("x3,0c0,0,255", b"") == "x3,0c0,0,255"       # explicit methods pass through


def test_level5_without_data_is_an_error_not_a_crash(cfg):
    import ctypes as C
    L = cfg.load()
    out = C.create_string_buffer(256)
    assert L.zpq_expand_method(None, b"5", None, 1000, out, 256) != 0
    assert L.zpq_expand_method(None, b"5", None, 0, out, 256) == 0 and out.value.startswith(b"x0,0w1i1")


def test_unpinned_preprocessors_are_refused(cfg):
    # strings that are no method at all (BWT + E8E9 above 16 MiB blocks was refused here until round 6: it has its post-processor
    # now, tests/test_pcomp_variants_cpu.py)
    for m in ("q1",):
        with pytest.raises(cfg.ConfigRefused):
            cfg.make_config(m)
    cfg.make_config("x5,7ci1")


@needs_ref
@pytest.mark.parametrize("method", ["x4,2,12,0,7,25,1c0,0,511i2", "x4,6,12,0,7,25,1c0,0,511i2", "x4,3ci1", "x4,7ci1", "x6,3ci1", "x4,4ci1,1,1,1,2a",
                                    "x0,2,4,0,7,21,1c0,0,511,1256i2"])
def test_level2_level3_configs_compile_like_the_reference_compiler(cfg, method):
    """Levels 2 / 3 / E8E9-only (served since round 3): the generated config source -- HCOMP with the LZ77 parse-state
    prelude, the post-processor programs -- goes through the REAL reference Compiler and must give our bytes."""
    src, args = cfg.make_config(method)
    assert cfg.compile_config(src, args) == ref_compiled(src, args)


METHODS = [LEVEL5, "x4,1,5,0,3,24", "x6,1,4,0,2,26", "x7,5,6,0,3,27", "x2,0ci1,1,1,1,2am", "x2,0ci1,1,1,1,2awm", "x0,0w2c0,1010,255i1c256ci1,1,1,1,1,1,2ac0,0,1009,255i1c0,10i1c0,2,0,255i1mm16ts19t0",
           "x1,0c0,0,511i2", "x3,0c1003,0,255,1002,240c256,1040s8,16,255m12,32t16,40s", "x0,0w3,48,10,255,31,1a16,1,2mm20", "x4,0c0,1300,255,1256,1512,15i12,21", "x0,0"]


@needs_ref
@pytest.mark.parametrize("method", METHODS)
def test_compiler_equals_reference_on_generated_configs(cfg, method):
    src, args = cfg.make_config(method)
    assert cfg.compile_config(src, args) == ref_compiled(src, args)


HAND = [
    # long forms: IFL / IFNOTL / ELSEL, loops too long for a short jump, nested comments, mixed case, $N+M
    "comp 3 4 5 $2+1 0 hcomp a= 1 ifl " + "a++ " * 200 + "elsel " + "b-- " * 150 + "endif halt pcomp x y z ; "
    "DO a=*b (a (nested) comment) A+= 255 IFNOT out ENDIF b++ a=b a== $1+3 UNTIL halt end",
    "comp 0 0 0 0 0 hcomp do " + "a*= 3 a+= 7 d=a hashd " * 40 + "a> 100 while do " + "c++ " * 130 + "forever halt end",
    "comp 1 2 0 0 2 0 const 7 1 avg 0 0 128 hcomp a<>a? ".replace("a<>a? ", "") + "b<>a c<>a d<>a *b<>a *c<>a *d<>a a! b! *d! a=0 *c=0 a=r 3 d=r 255 r=a 9 "
    "a+=*d a-=c a*=b a/= 5 a%=*b a&=d a&~c a|= 1 a^=*c a<<=b a>>= 3 a==*d a<c a>*b jt -3 jf 5 jmp 0 lj 300 *b=*c *d=a a=a error halt post 0 end",
    "comp 0 0 0 0 0 hcomp if halt do a++ end",        # left open: accepted by the reference, offset stays 0
    "comp 9 16 0 0 1 0 icm 5 hcomp ifnotl a=0 " + "out " * 140 + "else a++ endif do a-- a> 0 ifl " + "hash " * 200 + "endif until halt end",
]


@needs_ref
@pytest.mark.parametrize("i", range(len(HAND)))
def test_compiler_equals_reference_on_hand_written_programs(cfg, i):
    args = [2, 7, 0, 0, 0, 0, 0, 0, 0]
    assert cfg.compile_config(HAND[i], args) == ref_compiled(HAND[i], args)


@needs_ref
@pytest.mark.parametrize("name", list(cmconfigs.ALL))
def test_compiler_equals_reference_on_test_models(cfg, name):
    src = cmconfigs.ALL[name]
    assert cfg.compile_config(src, [0] * 9) == ref_compiled(src, [0] * 9)


@needs_ref
@pytest.mark.parametrize("bad", ["comp 0 0 0 0 0 hcomp a= 256 halt end", "comp 0 0 0 0 1 0 foo 1 hcomp halt end",
                                 "comp 0 0 0 0 0 hcomp endif halt end", "comp 0 0 0 0 0 hcomp a=q halt end", "comp 0 0 0 0 0 hcomp halt"])
def test_compiler_rejects_what_the_reference_rejects(cfg, bad):
    with pytest.raises(RuntimeError):
        orc.ref_compile(bad, [0] * 9)
    with pytest.raises(cfg.ConfigRefused):
        cfg.compile_config(bad, [0] * 9)


@needs_ref
def test_reference_encoder_reproduces_level5_d_block():
    """The 9.4 MB d block of sha256.zpaq: reference Predictor/Encoder over PASS + plaintext == the fixture bytes."""
    import json
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    b = json.load(open(os.path.join(G, "blocks.json")))[1]
    blk = arc[b["offset"]:b["offset"] + b["size"]]
    plain = lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())
    _, header, p = block_header(blk)
    q = blk.index(b"\0", blk.index(b"\0", p + 1) + 1) + 2
    coded = orc.ref_cm_encode(header, b"\0" + plain)
    assert blk[q:q + len(coded)] == coded and blk[q + len(coded)] == 253


@needs_ref
@pytest.mark.parametrize("name", ["zsfx", "zsfx32"])
def test_reference_encoder_reproduces_level5_fixture(name):
    """Sanity of the golden vector itself: the reference Predictor/Encoder driven with the fixture's header over
    PASS + plaintext gives exactly the fixture's coded bytes (this is what the GPU encoder is held to in -m gpu)."""
    arc = open(os.path.join(G, name + ".zpaq"), "rb").read()
    plain = lzma.decompress(open(os.path.join(G, name + "_plain.xz"), "rb").read())
    _, header, p = block_header(arc)
    q = arc.index(b"\0", arc.index(b"\0", p + 1) + 1) + 2
    coded = orc.ref_cm_encode(header, b"\0" + plain)
    assert arc[q:q + len(coded)] == coded
    assert arc[q + len(coded)] == 253 and arc[-1] == 255


# ---------------------------------------------------------------------------------------------------------------------
# Compressor::startBlock(int level) (ZSFX/libzpaq.h:1346): libzpaq's built-in models.  The byte array `models[]` sits in
# the half of libzpaq.cpp the snapshot lacks; it is libzpaq 7.15's (public domain), held here as the expectation.  The
# engine keeps the models as min / mid / max.cfg SOURCE: the same bytes must come out of its compiler AND out of the
# reference Compiler, and the reference Predictor must code with them.
# ---------------------------------------------------------------------------------------------------------------------
def _u8(v):
    return bytes((x + 256) % 256 for x in v)


BUILTIN = {
    1: _u8([26, 0, 1, 2, 0, 0, 2, 3, 16, 8, 19, 0, 0, 96, 4, 28, 59, 10, 59, 112, 25, 10, 59, 10, 59, 112, 56, 0]),
    2: _u8([69, 0, 3, 3, 0, 0, 8, 3, 5, 8, 13, 0, 8, 17, 1, 8, 18, 2, 8, 18, 3, 8, 19, 4, 4, 22, 24, 7, 16, 0, 7, 24, -1, 0, 17, 104, 74, 4, 95, 1, 59, 112, 10, 25, 59,
            112, 10, 25, 59, 112, 10, 25, 59, 112, 10, 25, 59, 112, 10, 25, 59, 10, 59, 112, 25, 69, -49, 8, 112, 56, 0]),
    3: _u8([-60, 0, 5, 9, 0, 0, 22, 1, -96, 3, 5, 8, 13, 1, 8, 16, 2, 8, 18, 3, 8, 19, 4, 8, 19, 5, 8, 20, 6, 4, 22, 24, 3, 17, 8, 19, 9, 3, 13, 3, 13, 3, 13, 3, 14, 7, 16,
            0, 15, 24, -1, 7, 8, 0, 16, 10, -1, 6, 0, 15, 16, 24, 0, 9, 8, 17, 32, -1, 6, 8, 17, 18, 16, -1, 9, 16, 19, 32, -1, 6, 0, 19, 20, 16, 0, 0, 17, 104, 74, 4,
            95, 2, 59, 112, 10, 25, 59, 112, 10, 25, 59, 112, 10, 25, 59, 112, 10, 25, 59, 112, 10, 25, 59, 10, 59, 112, 10, 25, 59, 112, 10, 25, 69, -73, 32, -17, 64, 47,
            14, -25, 91, 47, 10, 25, 60, 26, 48, -122, -105, 20, 112, 63, 9, 70, -33, 0, 39, 3, 25, 112, 26, 52, 25, 25, 74, 10, 4, 59, 112, 25, 10, 4, 59, 112, 25, 10, 4,
            59, 112, 25, 65, -113, -44, 72, 4, 59, 112, 8, -113, -40, 8, 68, -81, 60, 60, 25, 69, -49, 9, 112, 25, 25, 25, 25, 25, 112, 56, 0]),
}


def builtin_model(cfg, level):
    import ctypes as C
    L = cfg.load()
    L.zpq_builtin_model.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_builtin_model_source.restype = C.c_char_p
    L.zpq_builtin_model_source.argtypes = [C.c_int]
    buf = C.create_string_buffer(512)
    n = C.c_size_t(0)
    rc = L.zpq_builtin_model(level, buf, 512, C.byref(n))
    src = L.zpq_builtin_model_source(level)
    return rc, buf.raw[: n.value], src.decode() if src else None


@pytest.mark.parametrize("level", [1, 2, 3])
def test_builtin_models_are_libzpaqs(cfg, level):
    rc, h, src = builtin_model(cfg, level)
    assert rc == 0 and h == BUILTIN[level]
    assert h[0] | h[1] << 8 == len(h) - 2 and (h[6], len(h)) == {1: (2, 28), 2: (8, 71), 3: (22, 198)}[level]
    assert cfg.compile_config(src, [0] * 9) == (BUILTIN[level], b"")


def test_builtin_model_levels_out_of_range(cfg):
    for level in (0, 4, -1):
        rc, h, src = builtin_model(cfg, level)
        assert rc != 0 and src is None


@needs_ref
@pytest.mark.parametrize("level", [1, 2, 3])
def test_builtin_models_through_the_reference_compiler_and_predictor(cfg, level):
    import datagen
    _, h, src = builtin_model(cfg, level)
    assert ref_compiled(src, [0] * 9) == (BUILTIN[level], b"")
    x = b"\0" + datagen.text_like(6000, level) + datagen.binary_like(3000, level + 10)
    coded = orc.ref_cm_encode(h, x)
    assert len(coded) < len(x) * 3 // 4                     # it is a model, not noise
    assert orc.ref_cm_decode(h, coded, len(x) + 16) == x


def test_lz77_context_preamble_constant_follows_the_compiled_program():
    """The HCOMP of the byte-aligned LZ77 methods skips the post-processor section of the coded stream before it starts to
    track codes (ZPAQ's constant 111 = 3 + the 108 bytes of its level-2 program).  Here the program is restated (and differs
    with the E8E9 stage: 164 bytes since round 6), so the constant must be 3 + whatever THIS compiler makes of THIS
    program -- an edit of the program that forgot the constant would silently shift the model's parse state."""
    import re
    from zpaqfranz_amd import engine
    for method, want in (("x4,2,12,0,7,25,1c0,0,511i2", 108), ("x4,6,12,0,7,25,1c0,0,511i2", 164)):
        src, args = engine.make_config(engine.expand_method(method, b""))
        _, pcomp = engine.compile_config(src, args)
        skip = int(re.search(r"a=r 1 a== 0 if\s+a= (\d+)", src).group(1))
        assert len(pcomp) == want and skip == 3 + len(pcomp)
