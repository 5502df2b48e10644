"""Rows a11-a14 through the run-time specialised context-mixing coder (cm_jit.hip + cm_spec_src.inc): the kernels hiprtc
builds for a block header must code exactly like the reference Predictor + ZPAQL interpreter compiled in place
(oracle/_ref: ZSFX/libzpaq.cpp:1846-2058, :1033-1254) and like the engine's own generic kernels."""
import json
import lzma
import os

import pytest

import cmconfigs
import datagen
import orc

pytestmark = pytest.mark.gpu
G = orc.GOLDEN


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _hdr(name):
    return orc.ref_compile(cmconfigs.ALL[name], [0] * 9)[0]


def _method_header(method, data=b""):
    from zpaqfranz_amd import engine
    src, args = engine.make_config(engine.expand_method(method, data))
    return engine.compile_config(src, args)[0]


def test_specialised_kernel_equals_generic_kernel(eng, monkeypatch):
    """Same inputs through both device paths (ZPQ_CM_GENERIC=1 selects the interpreter-driven wave kernel)."""
    for name in cmconfigs.ALL:
        h = _hdr(name)
        inputs = [b"\0" + datagen.text_like(3000, 11), b"\0" + datagen.binary_like(2500, 12), b"", b"\0"]
        caps = [len(x) * 2 + 64 for x in inputs]
        spec = eng.cm_code([h] * len(inputs), inputs, caps, encode=True)
        monkeypatch.setenv("ZPQ_CM_GENERIC", "1")
        gen = eng.cm_code([h] * len(inputs), inputs, caps, encode=True)
        monkeypatch.delenv("ZPQ_CM_GENERIC")
        assert spec == gen, name
        assert [s for s, _ in spec] == [0] * len(inputs)


def test_segments_of_one_block_continue_the_model(eng, monkeypatch):
    """Compressor::startSegment ... endSegment more than once before endBlock / Decompresser::decompress after a second
    findFilename (ZSFX/libzpaq.cpp:2307-2337): the predictor and the HCOMP machine carry on from segment to segment, only the
    arithmetic coder starts afresh.  Measure: the real Predictor / Decoder kept across the segments (oracle/_ref).  Through
    the specialised kernels, the generic wave kernel and the one-lane kernel; an empty segment in the middle and at the end."""
    segs = [b"\0" + datagen.text_like(2500, 21), datagen.text_like(1800, 22), b"", datagen.binary_like(1500, 23), datagen.text_like(700, 21), b""]
    for name in ("mid", "alltypes", "order1_cm"):
        h = _hdr(name)
        want = orc.ref_cm_encode_segments(h, segs)
        assert want[1] != orc.ref_cm_encode(h, segs[1]), name                # the second segment does depend on the first
        assert orc.ref_cm_decode_segments(h, want, sum(map(len, segs)) + 16) == segs
        for env in ({}, {"ZPQ_CM_GENERIC": "1"}, {"ZPQ_CM_GENERIC": "1", "ZPQ_CM_ONE_LANE": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            st, got = eng.cm_code_segments(h, segs, sum(map(len, segs)) * 2 + 4096, encode=True)
            assert st == 0 and got == want, (name, env)
            st, back = eng.cm_code_segments(h, want, sum(map(len, segs)) + 16, encode=False)
            assert st == 0 and back == segs, (name, env)
            for k in env:
                monkeypatch.delenv(k)
    # a block of segments next to ordinary blocks in one call, and a coded segment that does not end where its length says
    h = _hdr("mid")
    want = orc.ref_cm_encode_segments(h, segs)
    st, back = eng.cm_code_segments(h, [want[0] + want[1][:5], want[1][5:]] + want[2:], sum(map(len, segs)) + 16, encode=False)
    assert st != 0


def test_many_blocks_several_headers_one_call(eng):
    """The queue: more blocks than waves in a workgroup, three headers in one call, ragged lengths (incl. empty)."""
    hs = [_hdr("mid"), _hdr("alltypes"), _hdr("order1_cm")]
    inputs, headers = [], []
    for k in range(150):
        n = [0, 1, 7, 300, 1200, 2500][k % 6] + k
        gen = datagen.text_like if k % 2 else datagen.mixed
        inputs.append(b"\0" + gen(n, 100 + k) if n else b"")
        headers.append(hs[k % 3])
    got = eng.cm_code(headers, inputs, [len(x) * 2 + 64 for x in inputs], encode=True)
    assert all(st == 0 for st, _ in got)
    for k in range(0, 150, 7):
        assert got[k][1] == orc.ref_cm_encode(headers[k], inputs[k]), k
    back = eng.cm_code(headers, [g for _, g in got], [len(x) + 16 for x in inputs], encode=False)
    for k, (st, b) in enumerate(back):
        assert st == 0 and b == inputs[k], k


@pytest.mark.parametrize("method", ["44", "54", "x0,0ci1"])
def test_methods_3_4_5_models_equal_reference(eng, method):
    """The models compressBlock's levels 3..5 select (the BWT model ci1 of 3, the order-1..6 mix of 4, the 22+ component
    chain of 5 with its 171-byte HCOMP), coded by the specialised kernels."""
    data = datagen.text_like(40000, 5) + datagen.binary_like(20000, 6)
    h = _method_header(method, data)
    x = b"\0" + data
    (st, got), = eng.cm_code([h], [x], [len(x) + 4096], encode=True)
    assert st == 0 and got == orc.ref_cm_encode(h, x)
    (st, back), = eng.cm_code([h], [got], [len(x) + 16], encode=False)
    assert st == 0 and back == x


def _raw_header(hh, hm, comps, hcomp):
    body = bytes([hh, hm, 0, 0, len(comps)]) + b"".join(bytes(c) for c in comps) + b"\0" + bytes(hcomp) + b"\0"
    return bytes([len(body) & 255, len(body) >> 8]) + body


def test_hcomp_jump_into_an_instruction_and_loops(eng):
    """ZPAQL lets a jump land inside another instruction's operand; the translation decodes from wherever control can
    go, like the interpreter.  First program: *d=a ; jmp +1 ; a= 56 -- the jump lands on the operand 56 = halt.
    Second: a counted loop (a= 3 ; a-- ; a> 0 ; jt -5 ; halt)."""
    x = b"\0" + datagen.text_like(1500, 3)
    for prog in ([112, 63, 1, 71, 56], [112, 71, 3, 2, 239, 0, 39, 251, 56]):
        h = _raw_header(2, 4, [(2, 16, 255)], prog)
        want = orc.ref_cm_encode(h, x)
        (st, got), = eng.cm_code([h], [x], [len(x) * 2 + 64], encode=True)
        assert st == 0 and got == want
        (st, back), = eng.cm_code([h], [got], [len(x) + 16], encode=False)
        assert st == 0 and back == x


def test_hcomp_endless_loop_is_stopped(eng):
    """jmp to itself: the reference would spin forever; the generated code counts backward jumps (2^24 free per byte + the
    block's one credit of 2^28) and refuses the block with ZPQ_ERR_LIMIT -- within seconds -- instead of hanging the GPU."""
    import time
    h = _raw_header(2, 4, [(2, 16, 255)], [63, 254])
    t0 = time.time()
    (st, _), = eng.cm_code([h], [b"\0abcdef"], [256], encode=True)
    assert st == -9 and (time.time() - t0 < 10 or os.environ.get("ZPQ_TEST_EMU"))
    # ... and the interpreter-driven kernel refuses it too (2^30 interpreted instructions; it used to end the call as if halted)
    if not os.environ.get("ZPQ_TEST_EMU"):
        os.environ["ZPQ_CM_GENERIC"] = "1"
        try:
            (st, _), = eng.cm_code([h], [b"\0a"], [256], encode=True)
        finally:
            del os.environ["ZPQ_CM_GENERIC"]
        assert st == -9


INIT_LOOP_CFG = """comp 1 0 0 0 1
  0 icm 8
hcomp
  c=a a=r 0 a== 0 if
    b=0 do
      a=0 do a++ a== 250 until
      b++ a=b a== 20
    until
    a= 1 r=a 0
  endif
  a=c a<<= 9 *d=a
  halt
post
  0
end
"""


def test_a_long_initialisation_loop_draws_on_the_blocks_credit(eng, monkeypatch):
    """VERDICT round 5, item 5 / ADVICE: a VALID HCOMP program may loop longer in one call than the per-byte limit of the generated
    code (one that initialises H or M on its first byte).  Such a call draws on a credit the block has once; with small limits
    (100 free backward jumps per byte, a credit of 6000) the 5020 jumps of this program's first call are coded exactly as the
    reference Predictor + ZPAQL interpreter code them, both directions; with a credit of 1000 the block is refused as
    ZPQ_ERR_LIMIT, not as malformed."""
    data = [b"\0" + datagen.text_like(1500, 71), b"\0" + datagen.binary_like(900, 72)]
    caps = [len(d) * 2 + 64 for d in data]
    h = orc.ref_compile(INIT_LOOP_CFG, [0] * 9)[0]
    want = [orc.ref_cm_encode(h, d) for d in data]
    assert eng.cm_code([h] * 2, data, caps, encode=True) == [(0, w) for w in want]          # (5020 < 2^24: no credit needed)
    monkeypatch.setenv("ZPQ_JIT_NOCACHE", "1")
    monkeypatch.setenv("ZPQ_JIT_HCOMP_GUARD", "100")
    monkeypatch.setenv("ZPQ_JIT_HCOMP_CREDIT", "6000")
    got = eng.cm_code([h] * 2, data, caps, encode=True)
    assert got == [(0, w) for w in want]
    assert eng.cm_code([h] * 2, want, [len(d) + 16 for d in data], encode=False) == [(0, d) for d in data]
    monkeypatch.setenv("ZPQ_JIT_HCOMP_CREDIT", "1000")
    assert [st for st, _ in eng.cm_code([h] * 2, data, caps, encode=True)] == [-9, -9]


def _fixture_dblock():
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    blk = json.load(open(os.path.join(G, "blocks.json")))[1]
    raw = arc[blk["offset"]: blk["offset"] + blk["size"]]
    hs = 13 + 5
    hsize = raw[hs] | raw[hs + 1] << 8
    header = raw[hs: hs + 2 + hsize]
    p = hs + 2 + hsize + 1
    p = raw.index(b"\0", p) + 1      # filename
    p = raw.index(b"\0", p) + 1      # comment
    p += 1                           # reserved
    return arc, header, raw[p: len(raw) - 22]     # ... 00 00 00 00 | fd sha1[20] ff


def _fixture_attributes(e_for_attrs=None):
    """{file name: (XXHASH64 hex, CRC-32 hex)} from the i blocks of the fixture (layout: SURVEY.md Appendix B.4)."""
    out = {}
    for k in (1, 2, 3):
        b = open(os.path.join(G, "iblock%d.bin" % k), "rb").read()
        p = 0
        while p + 9 <= len(b):
            date = int.from_bytes(b[p:p + 8], "little"); p += 8
            z = b.index(b"\0", p); name = b[p:z].decode(); p = z + 1
            if not date:
                continue
            na = int.from_bytes(b[p:p + 4], "little"); p += 4
            attr = b[p:p + na]; p += na
            ni = int.from_bytes(b[p:p + 4], "little"); p += 4 + 4 * ni
            if na >= 58:
                out[name] = (attr[16:32].decode(), attr[49:57].decode())
    return out


def test_reference_archive_in_full_both_directions():
    """The reference's own interoperability fixture (AUTOTEST/README.txt:1-40): AUTOTEST/sha256.zpaq, written by
    zpaqfranz -m5 on Windows.  (a) Jidac extract on the GPU returns 256 files whose SHA-256 are their names -- the
    reference's autotest criterion -- and `t` verifies every stored XXHASH64 / CRC-32; (b) the d block's 9 473 560
    bytes of plaintext code to exactly the archive's 121 236 bytes (23 components, 171-byte HCOMP).  The two
    directions run side by side on two contexts."""
    import hashlib
    import threading
    from zpaqfranz_amd import Engine, engine as E
    arc, header, coded = _fixture_dblock()
    plain = lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())
    assert header[6] == 23 and len(coded) == 121236 + 4 and coded[-4:] == b"\0\0\0\0" and len(plain) == 9473560   # + the four 0 bytes after the end-of-segment symbol
    res = {}

    def encode():
        e = Engine(0)
        try:
            res["enc"] = e.cm_code([header], [b"\0" + plain], [len(coded) + 4096], encode=True)[0]
        finally:
            e.close()

    def extract():
        e = Engine(0)
        try:
            res["files"] = E.jidac_extract(e, arc)
        finally:
            e.close()

    ts = [threading.Thread(target=encode), threading.Thread(target=extract)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st, got = res["enc"]
    assert st == 0 and got == coded
    files = res["files"]
    assert len(files) == 256
    for name, data in files.items():
        assert len(data) == 37000
        assert hashlib.sha256(data).hexdigest().upper() == os.path.basename(name).upper()[:64], name
    # the XXHASH64 / CRC-32 the reference stored for every file (what `t` compares; zpqj_verify does the same on the
    # device -- tests/test_gpu_verify.py -- and would decode the d block a second time here)
    import zlib
    attrs = _fixture_attributes(e_for_attrs=None)
    assert len(attrs) == 256
    for name, data in files.items():
        xx, crc = attrs[name]
        assert "%08X" % zlib.crc32(data) == crc and orc.xxh64(data) == int(xx, 16), name


@pytest.mark.parametrize("level", [1, 2, 3])
def test_builtin_models_of_startblock_level_equal_reference(eng, level):
    """Compressor::startBlock(int level) (ZSFX/libzpaq.h:1346): min / mid / max.cfg as zpq_builtin_model gives them, coded by
    the specialised kernels like the reference Predictor does."""
    import ctypes as C
    from zpaqfranz_amd import engine
    L = engine.load()
    L.zpq_builtin_model.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(512)
    n = C.c_size_t(0)
    assert L.zpq_builtin_model(level, buf, 512, C.byref(n)) == 0
    h = buf.raw[: n.value]
    x = b"\0" + datagen.text_like(30000, 40 + level) + datagen.binary_like(12000, 50 + level)
    (st, got), = eng.cm_code([h], [x], [len(x) + 4096], encode=True)
    assert st == 0 and got == orc.ref_cm_encode(h, x)
    (st, back), = eng.cm_code([h], [got], [len(x) + 16], encode=False)
    assert st == 0 and back == x
