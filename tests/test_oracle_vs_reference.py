"""Cross-checks the restated oracle against the REAL reference compiled in place
(oracle/_ref/libzpaqref.so <- /root/reference/ZSFX/libzpaq.cpp).  Skipped where that build is absent."""
import numpy as np
import pytest

import datagen
import orc

pytestmark = pytest.mark.ref

# every LZ77 level-1 argument set compressBlock emits for method "1x" (SURVEY.md Appendix C.3),
# at block sizes arg0 = 0 (1 MiB), 2 and 4 (16 MiB: the -m1 default "14")
def _argsets():
    out = []
    for a0 in (0, 2, 4):
        htsz = 19 + a0 + (a0 <= 6)
        out += [[a0, 1, 4, 0, 1, 15], [a0, 1, 4, 0, 2, 16], [a0, 1, 4, 0, 2, htsz], [a0, 1, 5, 0, 3, htsz], [a0, 1, 6, 0, 3, htsz]]
    return out


INPUTS = {
    "empty": b"",
    "one": b"a",
    "short": b"abcabcabcabcabcabcabc",
    "zeros": bytes(70000),
    "text": datagen.text_like(300000, 1),
    "binary": datagen.binary_like(300000, 2),
    "mixed": datagen.mixed(400000, 3),
    "random": datagen.random_bytes(100000, 4),
    "runs": (b"ab" * 3000 + b"x" + b"ab" * 30000 + bytes(range(256)) * 40),
}


@pytest.mark.parametrize("name", list(INPUTS))
def test_sha1_sha256(name):
    b = INPUTS[name]
    assert orc.sha1(b) == orc.ref_sha1(b)
    assert orc.sha256(b) == orc.ref_sha256(b)


@pytest.mark.parametrize("n", [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 128, 1000])
def test_sha_padding_edges(n):
    b = datagen.random_bytes(n, n)
    assert orc.sha1(b) == orc.ref_sha1(b)
    assert orc.sha256(b) == orc.ref_sha256(b)


@pytest.mark.parametrize("args", _argsets(), ids=lambda a: ",".join(map(str, a)))
@pytest.mark.parametrize("name", list(INPUTS))
def test_lz77_stream_identical_to_lzbuffer(name, args):
    b = INPUTS[name]
    ours = orc.lz77_encode(b, args)
    assert ours == orc.ref_lzbuffer(b, args)
    assert orc.lz77_decode(ours, len(b), rb=0) == b


def test_lz77_long_match_and_literal_limits():
    # maxMatch = 49152 and maxLiteral = 4096 (ZSFX/libzpaq.cpp:6275-6276)
    rng = np.random.default_rng(9)
    unit = rng.integers(0, 256, size=60000, dtype=np.uint8).tobytes()
    b = unit + unit + unit[:100] + datagen.random_bytes(20000, 5) + unit
    for args in ([0, 1, 5, 0, 3, 20], [4, 1, 5, 0, 3, 24]):
        ours = orc.lz77_encode(b, args)
        assert ours == orc.ref_lzbuffer(b, args)
        assert orc.lz77_decode(ours, len(b)) == b


L2_HASH_ARGS = [(4, 2, 5, 0, 3, 22), (4, 2, 12, 0, 3, 24), (0, 2, 4, 0, 1, 18), (5, 2, 6, 0, 2, 20), (4, 2, 4, 0, 0, 16)]


@pytest.mark.parametrize("args", L2_HASH_ARGS, ids=lambda a: ",".join(map(str, a)))
def test_lz77_level2_hash_finder_stream_identical_to_lzbuffer(args):
    """Byte-aligned codes from the hash-table finder (round 4): the restatement against the real LZBuffer, with matches of
    minMatch .. minMatch + 2 bytes at distances beyond 2^16, where the 'a far match must be longer' rule (:6415-6416) decides,
    a match beyond maxMatch (cut into pieces) and a literal run beyond maxLiteral."""
    rng = np.random.default_rng(21)
    base = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    far = bytearray(base)
    for i in range(600):
        p = int(rng.integers(0, 60000))
        far += base[p:p + args[2] + (i % 3)] + bytes(rng.integers(0, 256, size=3, dtype=np.uint8))
    for b in list(INPUTS.values()) + [bytes(far), base + base + datagen.random_bytes(6000, 3) + base[:100]]:
        assert orc.lz77_encode(b, args) == orc.ref_lzbuffer(b, args), (args, len(b))


# (block bits, level, minMatch, minMatch2, log2 bucket, log2 table, lookahead)
SECOND_CONTEXT_ARGS = [(4, 1, 4, 8, 3, 24, 0), (4, 1, 4, 8, 3, 24, 1), (4, 1, 5, 12, 2, 22, 2), (0, 1, 4, 6, 0, 18, 0), (4, 2, 4, 8, 3, 22, 1),
                       (4, 2, 6, 10, 1, 20, 3), (5, 1, 4, 9, 3, 25, 1), (4, 1, 4, 0, 3, 24, 2), (4, 1, 6, 3, 2, 20, 0)]


@pytest.mark.parametrize("args", SECOND_CONTEXT_ARGS, ids=lambda a: ",".join(map(str, a)))
def test_lz77_second_context_and_lookahead_identical_to_lzbuffer(args):
    """The restatement of LZBuffer's second, higher-order context and lookahead (round 5; ZSFX/libzpaq.cpp:6263-6290, 6373-6393,
    6411-6447) against the real LZBuffer: matches counted from the lookahead on and extended backwards, leading literals, both
    hashes in one table, levels 1 and 2 -- incl. repeats whose first bytes were changed (matches that start behind a lookahead)."""
    rng = np.random.default_rng(33)
    base = rng.integers(0, 256, size=50000, dtype=np.uint8).tobytes()
    rep = bytearray()
    for i in range(900):
        q = int(rng.integers(0, 49000))
        piece = bytearray(base[q:q + int(rng.integers(6, 60))])
        if i % 3 == 0 and len(piece) > 3:
            piece[0] ^= 0x5a
        if i % 5 == 0 and len(piece) > 4:
            piece[1] ^= 0x33
        rep += piece + bytes(rng.integers(0, 256, size=int(rng.integers(0, 4)), dtype=np.uint8))
    for b in list(INPUTS.values()) + [base + bytes(rep), bytes(20000), b"abcabcabc" * 30]:
        assert orc.lz77_encode(b, list(args)) == orc.ref_lzbuffer(b, list(args)), (args, len(b))


def test_e8e9_forward_and_inverse():
    rng = np.random.default_rng(11)
    b = bytearray(rng.integers(0, 256, size=200000, dtype=np.uint8).tobytes())
    for i in range(0, len(b) - 8, 7):      # dense, overlapping E8/E9 patterns
        b[i] = 0xE8 + (i & 1)
        b[i + 4] = 0xFF if i & 2 else 0
    b = bytes(b)
    f = orc.e8e9(b)
    assert f == orc.ref_e8e9(b)
    assert orc.e8e9_inverse(f) == b


@pytest.mark.parametrize("method", ["0", "1", "14,220,0", "14,100,0", "14,30,0", "14,20,1", "14,255,1", "10,5,0", "x0,0"])
@pytest.mark.parametrize("name", ["empty", "short", "text", "binary", "zeros"])
def test_blocks_decode_with_reference_decompresser(name, method):
    """Every block the oracle frames must be a valid ZPAQ block for the reference Decompresser
    (ZSFX/libzpaq.cpp:2239-2366), restore the input and carry a matching SHA-1."""
    b = INPUTS[name]
    blk, args = orc.compress_block(b, method, "jDC20240101000000d0000000001", "jDC\x01", True)
    r = orc.ref_decompress_block(blk, len(b) + 16)
    assert r["data"] == b and r["sha1_ok"] == 1 and r["consumed"] == len(blk)
    assert r["filename"] == b"jDC20240101000000d0000000001"
    assert r["comment"] == b"%d jDC\x01" % len(b)
    mine, meta = orc.decompress_block(blk, len(b) + 16)
    assert mine == b and meta[1] == 1
