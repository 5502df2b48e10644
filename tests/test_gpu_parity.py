"""GPU parity tests: every result of the HIP engine (through the C ABI, include/zpaqhip.h) must be
bit-identical to the CPU oracle (oracle/zpaq_oracle.cpp, itself pinned to the reference) and to the
reference's golden fixture.  Run on the MI355X box: python -m pytest tests -m gpu"""
import ctypes as C
import json
import lzma
import os
import struct

import numpy as np
import pytest

import datagen
import orc

pytestmark = pytest.mark.gpu
G = orc.GOLDEN
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def dplain():
    return lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())


# ---------------------------------------------------------------------------------------------------
# rows a2 / a18: SHA-1, SHA-256
# ---------------------------------------------------------------------------------------------------
def test_sha_padding_edges(eng):
    bufs = [datagen.random_bytes(n, n + 1) for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1000, 4096, 65537)]
    assert eng.sha1_many(bufs) == [orc.sha1(b) for b in bufs]
    assert eng.sha256_many(bufs) == [orc.sha256(b) for b in bufs]


def test_sha_many_ragged(eng):
    rng = np.random.default_rng(5)
    bufs = [datagen.random_bytes(int(rng.integers(0, 300000)), 100 + i) for i in range(700)]
    assert eng.sha1_many(bufs) == [orc.sha1(b) for b in bufs]
    assert eng.sha256_many(bufs[:200]) == [orc.sha256(b) for b in bufs[:200]]


def test_sha_more_extents_than_lanes_longest_first(eng):
    """Above ~131k extents the kernels deal extents longest-first (counting sort on the length class);
    every digest must still land in its own slot.  Overlapping extents of one buffer, ragged lengths."""
    import hashlib
    rng = np.random.default_rng(77)
    n = 150000
    data = datagen.random_bytes(6 << 20, 78)
    off = rng.integers(0, (6 << 20) - 70000, n)
    ln = rng.integers(0, 200, n)
    ln[::1000] = rng.integers(4096, 70000, len(ln[::1000]))       # a few long ones, so that the order matters
    bufs = [data[int(o):int(o) + int(l)] for o, l in zip(off, ln)]
    assert eng.sha1_many(bufs) == [hashlib.sha1(b).digest() for b in bufs]
    assert eng.sha256_many(bufs) == [hashlib.sha256(b).digest() for b in bufs]


def test_sha256_fixture_known_answers(eng, dplain):
    """AUTOTEST/README.txt:42-297: 256 files of 37 000 bytes named by their SHA-256."""
    names = []
    for k in (1, 2, 3):
        ib = open(os.path.join(G, "iblock%d.bin" % k), "rb").read()
        p = 0
        while p < len(ib):
            date = struct.unpack("<q", ib[p:p + 8])[0]; p += 8
            e = ib.index(b"\0", p); name = ib[p:e]; p = e + 1
            if date:
                na = struct.unpack("<I", ib[p:p + 4])[0]; p += 4 + na
                ni = struct.unpack("<I", ib[p:p + 4])[0]; p += 4 + 4 * ni
                names.append(os.path.basename(name.decode("latin1")).split(".")[0].lower())
    files = [dplain[i * 37000:(i + 1) * 37000] for i in range(256)]
    got = sorted(d.hex() for d in eng.sha256_many(files))
    assert got == sorted(names)


# ---------------------------------------------------------------------------------------------------
# row a1: fragmenter
# ---------------------------------------------------------------------------------------------------
def _oracle_frags(files, fragment=6, minf=4096, maxf=520192):
    out = []
    for fi, f in enumerate(files):
        off = 0
        for ln in orc.chunk(f, fragment, minf, maxf):
            out.append((fi, off, ln))
            off += ln
    return out


def test_fragmenter_fixture(eng, dplain):
    """All 388 fragment records of the golden h block (sizes AND SHA-1s), 256 files in one call."""
    h = open(os.path.join(G, "hblock_plain.bin"), "rb").read()
    want = [(h[4 + 24 * i: 24 + 24 * i], struct.unpack("<I", h[24 + 24 * i: 28 + 24 * i])[0]) for i in range(388)]
    files = [dplain[i * 37000:(i + 1) * 37000] for i in range(256)]
    frags = eng.fragment_files(files)
    assert [ln for _, _, ln in frags] == [u for _, u in want]
    digs = eng.sha1_many([files[f][o:o + ln] for f, o, ln in frags])
    assert digs == [s for s, _ in want]


def test_fragmenter_edge_cases(eng):
    files = [b"", b"a", bytes(5000), bytes(3 << 20), datagen.text_like(70000, 1), b"", datagen.random_bytes(4096, 2),
             datagen.random_bytes(4097, 3), b"ab" * 400000, datagen.binary_like(1 << 20, 4), b"x" * 4095,
             datagen.mixed((1 << 20) + 12345, 5), datagen.mixed(3 * (1 << 20), 6), datagen.text_like((1 << 20) - 1, 7),
             datagen.text_like(1 << 20, 8), datagen.random_bytes((2 << 20) + 1, 9)]
    assert eng.fragment_files(files) == _oracle_frags(files)


def test_fragmenter_multi_segment_file(eng):
    """A 24 MiB file spans 24 speculative segments: the stitcher must reproduce the serial chain."""
    f = datagen.mixed(24 << 20, 11)
    assert eng.fragment_files([f]) == _oracle_frags([f])


def test_fragmenter_lanes_pull_many_segments(eng, monkeypatch):
    """The speculative kernel is persistent: lanes pull segments from a device counter.  With the launch
    capped to one or two waves every lane walks several segments in turn (and their crossing
    fragments), over several files at once."""
    files = [datagen.mixed(40 << 20, 12), datagen.text_like((9 << 20) + 333, 13), b"", datagen.binary_like(5 << 20, 14),
             bytes(2 << 20), datagen.random_bytes((3 << 20) + 1, 15)]
    want = _oracle_frags(files)
    for waves, seg in (("1", "262144"), ("2", "65536"), ("1", "86016"), ("3", "1048576")):
        monkeypatch.setenv("ZPQ_FRAG_MAX_WAVES", waves)
        monkeypatch.setenv("ZPQ_FRAG_SEG", seg)          # the segment size is a per-call choice; results never depend on it
        assert eng.fragment_files(files) == want
    monkeypatch.delenv("ZPQ_FRAG_MAX_WAVES")
    monkeypatch.delenv("ZPQ_FRAG_SEG")
    assert eng.fragment_files(files) == want
    # long crossing walks are parked and resumed by a second launch: park nearly all of them / none of them
    for budget in ("4096", "70000", "0"):
        monkeypatch.setenv("ZPQ_FRAG_BUDGET", budget)
        assert eng.fragment_files(files) == want
    monkeypatch.setenv("ZPQ_FRAG_BUDGET", "1000")
    monkeypatch.setenv("ZPQ_FRAG_SEG", "65536")
    monkeypatch.setenv("ZPQ_FRAG_MAX_WAVES", "2")
    assert eng.fragment_files(files) == want


def test_fragmenter_periodic_and_constant_runs(eng):
    """Fully predictable data: the hash never forgets where its fragment began, so speculative and true
    chains never fall in step and the exact wave evaluator does the work (its steady-state fast path
    and the general path alternate at the edges of the runs)."""
    rng = np.random.default_rng(31)
    parts = []
    for i in range(40):
        kind = i % 5
        n = int(rng.integers(50000, 900000))
        if kind == 0: parts.append(bytes([int(rng.integers(0, 256))]) * n)
        elif kind == 1: parts.append((bytes(rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8)) * n)[:n])
        elif kind == 2: parts.append(datagen.text_like(n // 8, 100 + i))
        elif kind == 3: parts.append((b"ab" * n)[:n])
        else: parts.append(datagen.random_bytes(n // 16, 200 + i))
    big = b"".join(parts)
    files = [big, bytes(7 << 20), big[12345:3000000]]
    assert eng.fragment_files(files) == _oracle_frags(files)


def test_fragmenter_never_synchronising_input(eng):
    """Zeros cut only at MAX: speculative and true chains never meet (worst case, still exact)."""
    f = bytes(5 * (1 << 20) + 777)
    assert eng.fragment_files([f]) == _oracle_frags([f])


@pytest.mark.parametrize("fragment,minf,maxf", [(0, 64, 8128), (3, 512, 65024), (6, 4096, 520192), (8, 16384, 2080768)])
def test_fragmenter_other_fragment_settings(eng, fragment, minf, maxf):
    files = [datagen.mixed(3 << 20, 21), datagen.text_like(200000, 22), datagen.binary_like(1500000, 23)]
    p = eng.fragment_params(fragment, minf, maxf)
    assert eng.fragment_files(files, p) == _oracle_frags(files, fragment, minf, maxf)


# ---------------------------------------------------------------------------------------------------
# row a7: E8E9 pre-processor, against the REAL reference e8e9() (oracle/_ref) and the restatement
# ---------------------------------------------------------------------------------------------------
def test_e8e9_forward_equals_reference(eng):
    rng = np.random.default_rng(88)
    cases = [b"", b"\xe8", b"\xe8\0\0\0\0", b"\xe9\x01\x02\x03\xff", bytes([0xe8]) * 5000, b"\xe8\xff" * 3000, b"\xe8\x00\xe9\xff\xff" * 2000]
    # x86-like: opcodes every few bytes, high operand byte 00/ff often, chains of adjacent opcodes
    a = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    idx = rng.integers(0, len(a) - 8, 400000)
    a[idx] = np.where(rng.integers(0, 2, len(idx)) == 0, 0xe8, 0xe9)
    a[idx + 4] = np.where(rng.integers(0, 3, len(idx)) == 0, 0xff, 0x00)
    cases.append(a.tobytes())
    b = rng.integers(0, 4, 1 << 20, dtype=np.uint8)
    cases.append(bytes(np.choose(b, [0xe8, 0xe9, 0x00, 0xff]).astype(np.uint8)))      # dense chains
    cases.append(datagen.binary_like((1 << 20) + 3, 89))
    for c in cases:
        got = eng.e8e9(c)
        assert got == orc.e8e9(c)
        if orc.have_ref():
            assert got == orc.ref_e8e9(c)
        assert orc.e8e9_inverse(got) == c


# ---------------------------------------------------------------------------------------------------
# row a3: dedup
# ---------------------------------------------------------------------------------------------------
def test_dedup_first_occurrence(eng):
    rng = np.random.default_rng(3)
    uniq = [orc.sha1(struct.pack("<I", i)) for i in range(5000)]
    idx = rng.integers(0, 5000, size=60000)
    dig = b"".join(uniq[i] for i in idx)
    d_dig = eng.upload(dig)
    d_first = eng.alloc(4 * len(idx))
    eng.dedup_dev(d_dig.ptr, len(idx), d_first.ptr)
    eng.sync()
    got = struct.unpack("<%dI" % len(idx), d_first.download(4 * len(idx)))
    seen, want = {}, []
    for k, i in enumerate(idx):
        want.append(seen.setdefault(int(i), k))
    d_dig.free(); d_first.free()
    assert list(got) == want


# ---------------------------------------------------------------------------------------------------
# row a8: LZ77 level 1
# ---------------------------------------------------------------------------------------------------
LZ_INPUTS = {
    "empty": b"", "one": b"a", "tiny": b"abcabcabcabcabcabcabcabc", "nine": b"123456789",
    "zeros": bytes(200000), "text": datagen.text_like(400000, 1), "binary": datagen.binary_like(400000, 2),
    "mixed": datagen.mixed(600000, 3), "random": datagen.random_bytes(150000, 4),
    "runs": b"ab" * 3000 + b"x" + b"ab" * 30000 + bytes(range(256)) * 40,
}


def _argsets():
    out = []
    for a0 in (0, 4):
        htsz = 19 + a0 + (a0 <= 6)
        out += [[a0, 1, 4, 0, 1, 15], [a0, 1, 4, 0, 2, 16], [a0, 1, 4, 0, 2, htsz], [a0, 1, 5, 0, 3, htsz], [a0, 1, 6, 0, 3, htsz]]
    return out


@pytest.mark.parametrize("args", _argsets(), ids=lambda a: ",".join(map(str, a)))
def test_lz77_streams_bit_identical(eng, args):
    names = list(LZ_INPUTS)
    got = eng.lz77_encode([LZ_INPUTS[k] for k in names], [args] * len(names))
    for k, g in zip(names, got):
        want, trace = orc.lz77_encode(LZ_INPUTS[k], args, trace=True)
        assert len(g) == len(want), (k, len(g), len(want))
        assert g == want, k


L2_HASH_ARGS = [(4, 2, 5, 0, 3, 22), (4, 2, 12, 0, 3, 24), (0, 2, 4, 0, 1, 18), (5, 2, 6, 0, 2, 20), (4, 2, 4, 0, 0, 16)]
# (block bits, level, minMatch, minMatch2, log2 bucket, log2 table, lookahead): a second, higher-order context and lookahead
SECOND_CONTEXT_ARGS = [(4, 1, 4, 8, 3, 24, 0), (4, 1, 4, 8, 3, 24, 1), (4, 1, 5, 12, 2, 22, 2), (0, 1, 4, 6, 0, 18, 0), (4, 2, 4, 8, 3, 22, 1),
                       (4, 2, 6, 10, 1, 20, 3), (5, 1, 4, 9, 3, 25, 1), (4, 1, 4, 0, 3, 24, 2), (4, 1, 6, 3, 2, 20, 0)]


@pytest.mark.parametrize("args", SECOND_CONTEXT_ARGS)
def test_lz77_second_context_and_lookahead_equal_the_real_lzbuffer(eng, args):
    """args[3] = minMatch2 > 0 (and / or args[6] = lookahead > 0) with the hash-table finder: LZBuffer searches the bucket of a
    second, higher-order hash first -- matches counted from `lookahead` bytes on and extended backwards, the bytes in front become
    leading literals -- and both hashes share one table (ZSFX/libzpaq.cpp:6263-6290, 6373-6447).  The code stream of
    lz77_generic_kernel must be the REAL LZBuffer's, bit for bit (level 1) / byte for byte (level 2), on text, binary, mixed and
    constructed inputs, empty and tiny blocks, and the stream must decode back."""
    rng = np.random.default_rng(33)
    base = rng.integers(0, 256, size=50000, dtype=np.uint8).tobytes()
    rep = bytearray()
    for i in range(900):                       # repeats whose first bytes were changed: matches that start behind a lookahead
        q = int(rng.integers(0, 49000))
        piece = bytearray(base[q:q + int(rng.integers(6, 60))])
        if i % 3 == 0 and len(piece) > 3:
            piece[0] ^= 0x5a
        if i % 5 == 0 and len(piece) > 4:
            piece[1] ^= 0x33
        rep += piece + bytes(rng.integers(0, 256, size=int(rng.integers(0, 4)), dtype=np.uint8))
    blocks = [datagen.text_like(120000, 41), datagen.mixed(90000, 42), base + bytes(rep), datagen.binary_like(60000, 43), bytes(20000), b"", b"q",
              b"abcabcabc" * 30, datagen.text_like(9000, 44) * 3]
    out = eng.lz77_encode(blocks, [args] * len(blocks))
    for b, o in zip(blocks, out):
        assert o == orc.lz77_encode(b, list(args)), (args, len(b))             # the oracle's restatement (pinned by tests/test_oracle_vs_reference.py)
        if orc.have_ref():
            assert o == orc.ref_lzbuffer(b, list(args)), (args, len(b))        # the real LZBuffer
    if args[1] == 1:                           # and it is a stream the level-1 decoder restores
        for b, o in zip(blocks, out):
            assert orc.lz77_decode(o, len(b) + 64, rb=max(0, args[0] - 4)) == b


@pytest.mark.parametrize("args", L2_HASH_ARGS)
def test_lz77_level2_byte_codes_from_the_hash_table_finder(eng, args):
    """(args[1] & 3) == 2 with args[5] - args[0] < 21: LZBuffer's hash-table search (ZSFX/libzpaq.cpp:6373-6461) with the
    byte-aligned codes and their rule that a far match must be 1 / 2 bytes longer (:6415-6416, :6482-6547).  `far` holds
    matches of exactly minMatch .. minMatch + 2 bytes at distances beyond 2^16, where that rule decides."""
    rng = np.random.default_rng(21)
    base = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    far = bytearray(base)
    mm = args[2]
    for i in range(600):
        p = int(rng.integers(0, 60000))
        far += base[p:p + mm + (i % 3)] + bytes(rng.integers(0, 256, size=3, dtype=np.uint8))
    blocks = [datagen.text_like(200000, 31), datagen.mixed(150000, 32), bytes(far), datagen.binary_like(90000, 33), b"", b"z", b"ab" * 40]
    out = eng.lz77_encode(blocks, [args] * len(blocks))
    for b, o in zip(blocks, out):
        assert o == orc.lz77_encode(b, args), (args, len(b))
        if orc.have_ref():
            assert o == orc.ref_lzbuffer(b, args), (args, len(b))


def test_lz77_long_matches_and_literal_limit(eng):
    rng = np.random.default_rng(9)
    unit = rng.integers(0, 256, size=60000, dtype=np.uint8).tobytes()
    b = unit + unit + unit[:100] + datagen.random_bytes(20000, 5) + unit
    for args in ([0, 1, 5, 0, 3, 20], [4, 1, 5, 0, 3, 24]):
        assert eng.lz77_encode([b], [args])[0] == orc.lz77_encode(b, args)


def test_lz77_full_16mib_block(eng):
    """The -m1 default block: 2^24-4096 bytes, args 4,1,5,0,3,24 (hash table 2^24)."""
    b = datagen.mixed((1 << 24) - 4096, 77)
    args = [4, 1, 5, 0, 3, 24]
    got = eng.lz77_encode([b], [args])[0]
    assert got == orc.lz77_encode(b, args)
    st, back = eng.lz77_decode([got], [len(b)])[0]
    assert st == 0 and back == b


def test_lz77_decode(eng):
    names = list(LZ_INPUTS)
    streams = [orc.lz77_encode(LZ_INPUTS[k], [0, 1, 5, 0, 3, 20]) for k in names]
    res = eng.lz77_decode(streams, [len(LZ_INPUTS[k]) + 8 for k in names])
    for k, (st, data) in zip(names, res):
        assert st == 0 and data == LZ_INPUTS[k], k


def test_lz77_decode_fixture_iblocks(eng):
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    blocks = json.load(open(os.path.join(G, "blocks.json")))
    res = eng.decompress_blocks([arc[b["offset"]:b["offset"] + b["size"]] for b in blocks if b["filename"][17] != "d"],
                                [1 << 16] * 5)
    plains = [struct.pack("<q", blocks[1]["size"]), open(os.path.join(G, "hblock_plain.bin"), "rb").read()] + \
             [open(os.path.join(G, "iblock%d.bin" % k), "rb").read() for k in (1, 2, 3)]
    for r, want, b in zip(res, plains, [b for b in blocks if b["filename"][17] != "d"]):
        assert r["status"] == 0 and r["data"] == want and r["consumed"] == b["size"]
        assert r["sha1"].hex() == b["sha1"]


# ---------------------------------------------------------------------------------------------------
# rows a5 / a10 / a11: compressBlock framing
# ---------------------------------------------------------------------------------------------------
def test_compress_block_reproduces_fixture_blocks(eng):
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    blocks = json.load(open(os.path.join(G, "blocks.json")))
    plains = {0: struct.pack("<q", blocks[1]["size"]), 2: open(os.path.join(G, "hblock_plain.bin"), "rb").read(),
              3: open(os.path.join(G, "iblock1.bin"), "rb").read(), 4: open(os.path.join(G, "iblock2.bin"), "rb").read(),
              5: open(os.path.join(G, "iblock3.bin"), "rb").read()}
    ks = sorted(plains)
    res = eng.compress_blocks([plains[k] for k in ks], ["1" if blocks[k]["filename"][17] == "i" else "0" for k in ks],
                              [blocks[k]["filename"] for k in ks], ["jDC\x01"] * len(ks), True)
    for k, (st, out) in zip(ks, res):
        assert st == 0
        assert out == arc[blocks[k]["offset"]:blocks[k]["offset"] + blocks[k]["size"]], blocks[k]["filename"]


@pytest.mark.parametrize("method", ["0", "1", "14,220,0", "14,100,0", "14,30,0", "14,20,1", "14,255,1", "10,5,0", "x0,0", "x2,1,4,0,2,16"])
def test_compress_block_equals_oracle(eng, method):
    ins = [b"", b"abc", datagen.text_like(300000, 1), datagen.binary_like(200000, 2), bytes(100000), datagen.mixed(70000, 3)]
    fns = ["jDC20240101000000d%010d" % (i + 1) for i in range(len(ins))]
    res = eng.compress_blocks(ins, [method] * len(ins), fns, ["jDC\x01"] * len(ins), True)
    for b, fn, (st, out) in zip(ins, fns, res):
        want, _ = orc.compress_block(b, method, fn, "jDC\x01", True)
        assert st == 0 and out == want
    back = eng.decompress_blocks([o for _, o in res], [len(b) + 8 for b in ins])
    for b, r in zip(ins, back):
        assert r["status"] == 0 and r["data"] == b and r["sha1"] == orc.sha1(b)


def test_compress_block_without_sha1_and_comment(eng):
    b = datagen.text_like(50000, 9)
    (st, out), = eng.compress_blocks([b], ["1"], None, None, False)
    want, _ = orc.compress_block(b, "1", None, None, False)
    assert st == 0 and out == want


# ---------------------------------------------------------------------------------------------------
# rows a5/a6 + a11-a16 together: compressBlock with context-mixing methods.  The configuration comes from
# the host-side makeConfig/ZPAQL compiler (config.hip), the coding from the GPU Predictor/Encoder.
# ---------------------------------------------------------------------------------------------------
def test_compress_block_level5_reproduces_fixture_archive(eng):
    """ZSFX/zsfx32.zpaq is a single streaming block written with method "5": 23 components, no post-processor,
    comment = the size.  compressBlock over the plaintext must give the archive, byte for byte."""
    arc = open(os.path.join(G, "zsfx32.zpaq"), "rb").read()
    plain = lzma.decompress(open(os.path.join(G, "zsfx32_plain.xz"), "rb").read())
    (st, out), = eng.compress_blocks([plain], ["5"], [""], None, True)
    assert st == 0
    assert out == arc
    r, = eng.decompress_blocks([out], [len(plain) + 8])
    assert r["status"] == 0 and r["data"] == plain


def test_compress_block_level5_prefix_of_second_fixture(eng):
    """ZSFX/zsfx.zpaq (321 KB, same model): an arithmetic-coded stream is prefix-stable, so the first 48 KiB of the
    plaintext must code to the first bytes of the archive's stream (all but the few bytes the end-of-segment flush
    touches).  The whole archive is reproduced on the CPU side by the reference coder (tests/test_config_cpu.py)."""
    arc = open(os.path.join(G, "zsfx.zpaq"), "rb").read()
    plain = lzma.decompress(open(os.path.join(G, "zsfx_plain.xz"), "rb").read())[:48 << 10]
    method = "x0,0w1i1c256ci1,1,1,1,1,1,2ac0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0"     # what "5" expands to there
    (st, out), = eng.compress_blocks([plain], [method], [""], None, True)
    assert st == 0

    def coded(b):
        k = b.index(b"zPQ"); p = k + 7 + (b[k + 5] | b[k + 6] << 8)
        q = b.index(b"\0", b.index(b"\0", p + 1) + 1) + 2
        return b[:k + 7 + (b[k + 5] | b[k + 6] << 8)], b[q:]
    h1, c1 = coded(out)
    h2, c2 = coded(arc)
    assert h1 == h2
    stable = len(c1) - 22 - 4 - 8          # minus SHA-1 trailer, terminator and the flushed tail
    assert stable > 10000 and c1[:stable] == c2[:stable]


def _reference_cm_block(data, method, fn, comment, sha):
    """What compressBlock writes for a context-mixing method, with the coding done by the REAL reference
    Predictor/Encoder (oracle/_ref) and the configuration by the reference Compiler over the makeConfig source."""
    from zpaqfranz_amd import engine
    x = engine.expand_method(method, data)
    src, args = engine.make_config(x)
    header, pcomp = orc.ref_compile(src, args)
    assert header[6] > 0
    body = orc.lz77_encode(data, args[:6]) if args[1] == 1 else data
    pre = (b"\1" + pcomp) if pcomp else b"\0"            # ref pcomp already carries its 2-byte length
    coded = orc.ref_cm_encode(header, pre + body)
    cs = str(len(data)) + ((" " + comment) if comment else "")
    out = bytes.fromhex("376b5374a03183d38cb228b0d3") + b"zPQ\1\1" + header + b"\1" + (fn or "").encode("latin1") + b"\0" + \
        cs.encode("latin1") + b"\0\0" + coded
    return out + ((b"\xfd" + orc.sha1(data)) if sha else b"\xfe") + b"\xff"


@needs_ref
@pytest.mark.parametrize("method", ["4", "44,128,1", "x0,0w2c0,1010,255i1m", "x0,1,5,0,3,20c0,0,255i1", "x1,0c0,0,1004,255i1c0,5i1c256ac0,2,0,255mm16ts19t0"])
def test_compress_block_cm_methods_equal_reference_coder(eng, method):
    ins = [datagen.text_like(30000, 51), datagen.binary_like(24000, 52), b"", b"z", bytes(5000) + datagen.random_bytes(3000, 53)]
    fns = ["jDC20240101000000d%010d" % (i + 1) for i in range(len(ins))]
    res = eng.compress_blocks(ins, [method] * len(ins), fns, ["jDC\x01"] * len(ins), True)
    for b, fn, (st, out) in zip(ins, fns, res):
        assert st == 0
        assert out == _reference_cm_block(b, method, fn, "jDC\x01", True)
    back = eng.decompress_blocks([o for _, o in res], [len(b) + 8 for b in ins])
    for b, r in zip(ins, back):
        assert r["status"] == 0 and r["data"] == b and r["sha1"] == orc.sha1(b)


def test_unsupported_methods_are_refused_not_approximated(eng):
    # what is still outside the implemented family is refused, never approximated: pre-processor numbers that do not exist, a
    # secondary context beyond 64 bytes.  (BWT + E8E9 above 16 MiB blocks, refused until round 6, is served: test_gpu_m3.py.  Levels
    # 2 / 3 / E8E9-only are served since round 3, level 2 from the hash-table finder since round 4, a secondary LZ77 context
    # and lookahead since round 5: test_lz77_second_context_and_lookahead_equal_the_real_lzbuffer.)
    res = eng.compress_blocks([b"hello world" * 100] * 2, ["x4,1,4,80,3,24", "x4,9ci1"], None, None, True)
    assert [st for st, _ in res] == [-5, -5]
    (st, blk), = eng.compress_blocks([b"hello world" * 100], ["x4,1,4,8,3,24,1"], None, None, True)       # second context of 8 bytes, lookahead 1
    assert st == 0 and eng.decompress_blocks([blk], [2000])[0]["data"] == b"hello world" * 100
    (st, blk), = eng.compress_blocks([b"hello world" * 100], ["x4,6,4,0,3,24c0"], None, None, True)       # byte codes from the hash-table finder + E8E9
    assert st == 0 and eng.decompress_blocks([blk], [2000])[0]["data"] == b"hello world" * 100
    (st, blk), = eng.compress_blocks([b"hello world" * 100], ["14,100,2"], None, None, True)
    assert st == 0 and eng.decompress_blocks([blk], [2000])[0]["data"] == b"hello world" * 100


def test_corrupt_block_is_detected(eng):
    b = datagen.text_like(50000, 10)
    (st, out), = eng.compress_blocks([b], ["1"], ["f"], None, True)
    bad = bytearray(out); bad[len(bad) // 2] ^= 0x55
    r, = eng.decompress_blocks([bytes(bad)], [len(b) + 64])
    assert r["status"] != 0 or r["data"] != b


# ---------------------------------------------------------------------------------------------------
# the libzpaq-shaped C++ shim, driven like Jidac's worker threads drive libzpaq
# ---------------------------------------------------------------------------------------------------
def test_libzpaq_shim_multithreaded_cpp_caller(tmp_path):
    import subprocess
    from zpaqfranz_amd import build
    build.build(verbose=False)
    drv = build.build_shim_driver(str(tmp_path / "shim_driver"))
    data = datagen.mixed(5 * (1 << 20) + 12345, 31)
    (tmp_path / "in.bin").write_bytes(data)
    block = 1 << 20
    r = subprocess.run([drv, str(tmp_path / "in.bin"), "14,128,0", "4", str(block), str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    s1, s256 = lines[0].split()
    assert s1 == orc.sha1(data).hex() and s256 == orc.sha256(data).hex()
    assert lines[1].startswith("unsupported: refused")
    want = b""
    for i in range(0, len(data), block):
        blk, _ = orc.compress_block(data[i:i + block], "14,128,0", "jDC20240101000000d%010d" % (i // block + 1), "jDC\x01", True)
        want += blk
    assert (tmp_path / "out.zpaq").read_bytes() == want
    assert (tmp_path / "out.back").read_bytes() == data
    if orc.have_ref():     # and the REAL reference decoder accepts the archive the shim wrote
        assert orc.ref_decompress(want, len(data) + 64) == data


def test_libzpaq_shim_decompresser_class_reads_fixture_archives(tmp_path):
    """libzpaq::Decompresser of the shim, driven like decompressThread (ZSFX/zsfx.cpp:1783-1834), over the reference's
    own archives: names, sizes, output SHA-1 and the stored SHA-1 records must agree -- every block DECODED, the 9.4 MB
    context-mixing d block of AUTOTEST/sha256.zpaq included (the specialised coder: ~100 s; it was skipped while the
    coder ran at 10 KB/s) -- and zsfx32.zpaq (23 components) likewise."""
    import subprocess
    from zpaqfranz_amd import build
    build.build(verbose=False)
    drv = build.build_shim_driver(str(tmp_path / "shim_driver"))
    blocks = json.load(open(os.path.join(G, "blocks.json")))
    r = subprocess.run([drv, "--extract", os.path.join(G, "sha256.zpaq"), "100000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [l.split("|") for l in r.stdout.strip().splitlines()]
    assert len(lines) == len(blocks)
    for l, b in zip(lines, blocks):
        assert l[0] == b["filename"] and l[4] == "1" and l[5] == b["sha1"]
        assert l[2] == "decoded" and int(l[1]) == b["usize"] and l[3] == b["sha1"]
    r = subprocess.run([drv, "--extract", os.path.join(G, "zsfx32.zpaq"), "1000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    (l,) = [x.split("|") for x in r.stdout.strip().splitlines()]
    plain = lzma.decompress(open(os.path.join(G, "zsfx32_plain.xz"), "rb").read())
    assert l[0] == "" and int(l[1]) == len(plain) and l[3] == orc.sha1(plain).hex() == l[5] and float(l[6]) > 5e7


def test_fragmenter_fragments_longer_than_a_segment(eng):
    """Fragments that swallow whole 256 KiB speculation segments (no cut inside a segment): incompressible
    data with max-size fragments, and a run whose period defeats the rolling hash."""
    files = [datagen.random_bytes(3 << 20, 41), bytes(range(256)) * 9000, datagen.random_bytes((1 << 20) + 5, 42)]
    p = eng.fragment_params(2, 1 << 19, 2 << 20)        # min 512 KiB, max 2 MiB: every fragment spans >= 2 segments
    assert eng.fragment_files(files, p) == _oracle_frags(files, 2, 1 << 19, 2 << 20)
    assert eng.fragment_files(files) == _oracle_frags(files)


# ---------------------------------------------------------------------------------------------------
# rows a11-a16: context mixing (Predictor + arithmetic coder + ZPAQL HCOMP), checked against the
# REAL reference Predictor/Decoder compiled in place (oracle/_ref)
# ---------------------------------------------------------------------------------------------------
import cmconfigs


@needs_ref
@pytest.mark.parametrize("name", list(cmconfigs.ALL))
def test_cm_encode_decode_equal_reference(eng, name):
    header, _ = orc.ref_compile(cmconfigs.ALL[name], [0] * 9)
    inputs = [b"", b"\0", b"\0" + datagen.text_like(6000, 5), b"\0" + datagen.binary_like(5000, 6), b"\0" + bytes(3000),
              b"\0" + datagen.random_bytes(2000, 7)]
    want = [orc.ref_cm_encode(header, x) for x in inputs]
    got = eng.cm_code([header] * len(inputs), inputs, [len(x) + len(x) // 2 + 64 for x in inputs], encode=True)
    for x, w, (st, g) in zip(inputs, want, got):
        assert st == 0 and g == w, (name, len(x))
    back = eng.cm_code([header] * len(inputs), want, [len(x) + 16 for x in inputs], encode=False)
    for x, (st, g) in zip(inputs, back):
        assert st == 0 and g == x
        assert orc.ref_cm_decode(header, want[inputs.index(x)], len(x) + 16) == x


def test_cm_decode_fixture_dblock_prefix(eng, dplain):
    """The -m5 d block of AUTOTEST/sha256.zpaq (23 components, 171-byte HCOMP): decode the first bytes of its
    arithmetic-coded stream and compare with the golden plaintext (first decoded byte is the PASS marker 0)."""
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    blk = json.load(open(os.path.join(G, "blocks.json")))[1]
    raw = arc[blk["offset"]: blk["offset"] + blk["size"]]
    hs = 13 + 5
    hsize = raw[hs] | raw[hs + 1] << 8
    header = raw[hs: hs + 2 + hsize]
    assert header[6] == 23
    p = hs + 2 + hsize
    assert raw[p] == 1
    p += 1
    p = raw.index(b"\0", p) + 1      # filename
    p = raw.index(b"\0", p) + 1      # comment
    p += 1                           # reserved
    coded = raw[p: len(raw) - 22]    # ... 00 00 00 00 | fd sha1[20] ff
    assert coded[-4:] == b"\0\0\0\0"
    n = 3000
    (st, got), = eng.cm_code([header], [coded], [n + 1], encode=False)
    assert st == -4                  # stopped at out_cap: the stream holds 9 473 561 bytes
    assert got == b"\0" + dplain[:n]


@needs_ref
def test_generic_pcomp_vm_runs_the_lz77_program(eng):
    """Row a14/a16: the generic ZPAQL interpreter executing the 302-byte level-1 PCOMP byte by byte must give
    what the reference PostProcessor gives (and what the native LZ77 decoder gives)."""
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    blk = json.load(open(os.path.join(G, "blocks.json")))[3]
    raw = arc[blk["offset"]: blk["offset"] + blk["size"]]
    plain = open(os.path.join(G, "iblock1.bin"), "rb").read()
    k = raw.index(b"\x01\x2e\x01")
    pcomp = raw[k + 3: k + 3 + 302]
    stream = orc.lz77_encode(plain, [0, 1, 5, 0, 3, 20])
    assert eng.pcomp_run(pcomp, 0, 20, stream, len(plain) + 64) == plain
    assert orc.ref_postprocess(b"\x01\x2e\x01" + pcomp + stream, 0, 20, len(plain) + 64) == plain


def _frame(header, body, level, plain, fn=b"f", comment=b"c"):
    """Frames one ZPAQ block by hand (tag, zPQ, level, 1, header, segment, body, SHA-1 trailer, 255)."""
    tag = bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])
    return tag + b"zPQ" + bytes([level, 1]) + header + b"\x01" + fn + b"\0" + comment + b"\0\0" + body + b"\xfd" + orc.sha1(plain) + b"\xff"


@needs_ref
def test_decompress_generic_blocks(eng):
    """Decompresser for ANY single-segment block: context-model coded data (mid config) with and without a
    post-processor, and a stored block whose PCOMP is not the known LZ77 program (generic ZPAQL VM)."""
    plain = datagen.text_like(7000, 21)
    # 1. modelled, PASS
    hdr, _ = orc.ref_compile(cmconfigs.MID, [0] * 9)
    b1 = _frame(hdr, orc.ref_cm_encode(hdr, b"\0" + plain), 1, plain)
    # 2. modelled + PCOMP that undoes "every byte + 7"
    cfg = cmconfigs.ORDER1_CM.replace("end\n", "pcomp add7 ;\n  a> 255 if halt endif a-= 7 out halt\nend\n")
    hdr2, pc = orc.ref_compile(cfg, [0] * 9)
    assert len(pc) > 2
    enc_in = bytes([1]) + pc + bytes((c + 7) & 255 for c in plain)       # pc already carries its 2-byte length
    b2 = _frame(hdr2, orc.ref_cm_encode(hdr2, enc_in), 1, plain)
    # 3. stored (n = 0) + the same unknown PCOMP
    hdr3, pc3 = orc.ref_compile("comp 0 0 0 0 0\nhcomp\n halt\npcomp add7 ;\n  a> 255 if halt endif a-= 7 out halt\nend\n", [0] * 9)
    payload = bytes([1]) + pc3 + bytes((c + 7) & 255 for c in plain)
    b3 = _frame(hdr3, struct.pack(">I", len(payload)) + payload + b"\0\0\0\0", 2, plain)
    blocks = [b1, b2, b3]
    for b in blocks:      # the hand framing is right: the reference decoder accepts it
        r = orc.ref_decompress_block(b, len(plain) + 16)
        assert r["data"] == plain and r["sha1_ok"] == 1
    res = eng.decompress_blocks(blocks, [len(plain) + 16] * 3)
    for r, b in zip(res, blocks):
        assert r["status"] == 0 and r["data"] == plain and r["consumed"] == len(b) and r["sha1"] == orc.sha1(plain)


# ---------------------------------------------------------------------------------------------------
# row a19 / section 8f-1: whole journaling archives (add -> reference reads it; extract -> originals)
# ---------------------------------------------------------------------------------------------------
def _walk(arc):
    off, out = 0, []
    while off < len(arc):
        b = orc.ref_decompress_block(arc[off:], 40 << 20)
        out.append(b)
        off += b["consumed"]
    return out


def test_journaling_archive_add_and_extract(eng):
    from zpaqfranz_amd import engine as E
    shared = datagen.mixed(3 << 20, 51)
    files = [("docs/a.txt", datagen.text_like(900000, 52)), ("docs/b.bin", datagen.binary_like(700000, 53)), ("empty", b""),
             ("big/one", shared + datagen.text_like(200000, 54)), ("big/two", shared), ("z/tiny", b"hello")]
    v1, st1 = E.jidac_add(eng, b"", files, 20240101120000)
    assert st1["new_fragments"] < st1["fragments"]          # big/two repeats big/one's fragments
    got = E.jidac_extract(eng, v1)
    assert got == {n: d for n, d in files}
    # second version: one file changed, one added, everything else dedups against version 1
    files2 = list(files)
    files2[0] = ("docs/a.txt", files[0][1][:400000] + b"EDIT" + files[0][1][400000:])
    files2.append(("new/file", datagen.text_like(300000, 55)))
    v2, st2 = E.jidac_add(eng, v1, files2, 20240202120000)
    assert st2["unique_bytes"] < 0.2 * sum(len(d) for _, d in files2)
    arc = v1 + v2
    assert E.jidac_extract(eng, arc) == {n: d for n, d in files2}
    if orc.have_ref():
        # the REAL reference decoder walks the archive: every block decodes with a matching SHA-1, block
        # names follow jDC<date><type><num>, c blocks hold the d-block byte counts
        blocks = _walk(arc)
        kinds = "".join(chr(b["filename"][17]) for b in blocks)
        assert kinds.startswith("cd") and "h" in kinds and kinds.endswith("i")
        assert all(b["sha1_ok"] == 1 and b["comment"].endswith(b" jDC\x01") for b in blocks)
        for k, b in enumerate(blocks):
            if chr(b["filename"][17]) == "c":
                csize = struct.unpack("<q", b["data"])[0]
                dsum, j = 0, k + 1
                while j < len(blocks) and chr(blocks[j]["filename"][17]) == "d":
                    dsum += blocks[j]["consumed"]; j += 1
                assert csize == dsum


# ---------------------------------------------------------------------------------------------------
# row e: the multi-rank add path, end to end.  Two ranks (gloo collectives through the host, both on
# GPU 0 -- RCCL itself cannot put two ranks on one device) must produce exactly the d blocks a single
# process produces for the concatenated inputs: table all-gather, global dedup, ownership, seam exchange.
# ---------------------------------------------------------------------------------------------------
def test_two_rank_add_is_bit_identical_to_serial(tmp_path):
    # (two contexts per rank: phase_a of step i+1 runs on a helper thread beside phase_b of step i, every collective is
    # issued by the main thread; the archive dumped is the one of the last step, which ran on the second context)
    import subprocess, sys
    from zpaqfranz_amd import sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "two_rank.bin")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--copies", "2",
           "--scale", "0.05", "--dist-backend", "gloo", "--same-device", "--own-corpus", "--no-cpu-baseline", "--no-verify", "--dump-archive", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    got = open(out, "rb").read()
    # serial expectation with the CPU oracle
    files = []
    for rank in range(2):
        corpus = datagen.silesia_like(seed=rank, scale=0.05)
        for _ in range(2):
            files += [b for _, b in corpus]
    frags = []
    for f in files:
        off = 0
        for ln in orc.chunk(f):
            frags.append(f[off:off + ln]); off += ln
    seen, uniq = {}, []
    for fr in frags:
        d = orc.sha1(fr)
        if d not in seen:
            seen[d] = len(uniq); uniq.append(fr)
    blk, nb = sharding.pack_blocks(np.array([len(u) for u in uniq]))
    want = b""
    for b in range(nb):
        idx = np.nonzero(blk == b)[0]
        body = b"".join(uniq[i] for i in idx)
        body += b"".join(struct.pack("<I", len(uniq[i])) for i in idx) + struct.pack("<II", 0, len(idx))
        fb, _ = orc.compress_block(body, "14", "jDC20240101000000d%010d" % (int(idx[0]) + 1), "jDC\x01", True)
        want += fb
    assert len(got) == len(want)
    assert got == want


def test_two_rank_shared_corpus_equals_the_single_gpu_archive(tmp_path):
    """Strong scaling (BASELINE metric: one Silesia x N corpus on 1/2/4/8 GPUs): --shared-corpus splits ONE corpus by file
    range over the ranks, so every fragment of rank 1 is a duplicate of one on rank 0 and only the all-gathered fragment
    tables find that out.  The stitched d blocks must equal, byte for byte, what one GPU writes for the whole corpus."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one, two = str(tmp_path / "one.bin"), str(tmp_path / "two.bin")
    # (scale 0.2: three d blocks, so that the balanced ownership gives rank 1 a block whose fragments all live on rank 0)
    common = ["--steps", "1", "--warmup", "1", "--copies", "4", "--scale", "0.2", "--no-cpu-baseline", "--no-verify", "--workload", "silesia_x256_m1"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dump-archive", one] + common, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    # no launcher in front: `bench.py --gpus 2` starts its two ranks itself, and one corpus split over the ranks is the default
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--same-device", "--dump-archive", two] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    a, b = open(one, "rb").read(), open(two, "rb").read()
    assert len(a) > 100000 and a == b
