"""Pins the CPU oracle (oracle/zpaq_oracle.cpp) against the reference's golden archive
AUTOTEST/sha256.zpaq (copied to tests/golden by tests/golden/make_golden.py, which decodes it with
the real reference decoder).  SURVEY.md section 8c lists what each block pins."""
import hashlib
import json
import lzma
import os
import struct

import pytest

import orc

G = orc.GOLDEN
ARC = open(os.path.join(G, "sha256.zpaq"), "rb").read()
BLOCKS = json.load(open(os.path.join(G, "blocks.json")))


@pytest.fixture(scope="module")
def dplain():
    return lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())


def test_archive_digest():
    # AUTOTEST/README.txt:37
    assert hashlib.sha256(ARC).hexdigest().upper() == "D90223FAEE2878D7854B9438864B4856A3C1F920C34EFB8C136A8949B54E5400"
    assert orc.sha256(ARC).hex().upper() == "D90223FAEE2878D7854B9438864B4856A3C1F920C34EFB8C136A8949B54E5400"


def _htable():
    h = open(os.path.join(G, "hblock_plain.bin"), "rb").read()
    bsize = struct.unpack("<I", h[:4])[0]
    frags = [(h[4 + 24 * i: 24 + 24 * i], struct.unpack("<I", h[24 + 24 * i: 28 + 24 * i])[0]) for i in range((len(h) - 4) // 24)]
    return bsize, frags


def test_dblock_trailer_and_hblock(dplain):
    """d block = fragment bytes + usize[4]*n + 0[4] + n[4] (ZSFX/zsfx.cpp:1468-1500 reads it back)."""
    bsize, frags = _htable()
    assert len(frags) == 388
    assert bsize == BLOCKS[1]["size"]
    n = struct.unpack("<I", dplain[-4:])[0]
    assert n == 388 and dplain[-8:-4] == b"\0\0\0\0"
    sizes = struct.unpack("<%dI" % n, dplain[-8 - 4 * n:-8])
    assert list(sizes) == [u for _, u in frags]
    assert sum(sizes) + 4 * n + 8 == len(dplain)


def test_chunker_and_sha1_reproduce_hblock(dplain):
    """Row a1+a2: the fragmenter and SHA-1 reproduce all 388 {sha1, size} records.  The d block
    holds the 256 files in archive order; every file is 37 000 bytes (AUTOTEST/README.txt)."""
    _, frags = _htable()
    data = dplain[: 256 * 37000]
    got = []
    for f in range(256):
        fb = data[f * 37000:(f + 1) * 37000]
        off = 0
        for ln in orc.chunk(fb):
            got.append((orc.sha1(fb[off:off + ln]), ln))
            off += ln
        assert off == 37000
    assert got == frags


def test_sha256_known_answers(dplain):
    """Row a18: AUTOTEST/README.txt:42-297 -- each file is named by its SHA-256.  The i blocks carry
    the names; check every one of the 256 files."""
    names = []
    for k in (1, 2, 3):
        ib = open(os.path.join(G, "iblock%d.bin" % k), "rb").read()
        p = 0
        while p < len(ib):
            date = struct.unpack("<q", ib[p:p + 8])[0]; p += 8
            e = ib.index(b"\0", p); name = ib[p:e]; p = e + 1
            if date:
                na = struct.unpack("<I", ib[p:p + 4])[0]; p += 4 + na
                ni = struct.unpack("<I", ib[p:p + 4])[0]; p += 4
                ptrs = struct.unpack("<%dI" % ni, ib[p:p + 4 * ni]); p += 4 * ni
                names.append((name, ptrs))
    assert len(names) == 256
    _, frags = _htable()
    offs = [0]
    for _, u in frags:
        offs.append(offs[-1] + u)
    checked = 0
    for name, ptrs in names:
        body = b"".join(dplain[offs[q - 1]:offs[q]] for q in ptrs)   # 1-based fragment ids
        stem = os.path.basename(name.decode("latin1")).split(".")[0]
        if len(stem) == 64:
            assert orc.sha256(body).hex().upper() == stem.upper()
            assert hashlib.sha256(body).hexdigest().upper() == stem.upper()
            checked += 1
    assert checked == 256


@pytest.mark.parametrize("k", [0, 2, 3, 4, 5])
def test_compress_block_reproduces_fixture_blocks(k):
    """Rows a5/a8/a10/a11 (stored mode): recompressing the plaintext of the c, h and i blocks with
    methods "0" / "1" reproduces the archive bytes exactly (framing, header, 302-byte PCOMP,
    LZ77 level-1 stream x0,1,5,0,3,20, SHA-1 trailer)."""
    b = BLOCKS[k]
    raw = ARC[b["offset"]: b["offset"] + b["size"]]
    plain, meta = orc.decompress_block(raw, 1 << 20)
    assert meta[0] == b["size"] and meta[1] == 1
    kind = b["filename"][17]
    if kind == "c":
        assert plain == struct.pack("<q", BLOCKS[1]["size"])
    if kind == "h":
        assert plain == open(os.path.join(G, "hblock_plain.bin"), "rb").read()
    if kind == "i":
        assert plain == open(os.path.join(G, "iblock%d.bin" % int(b["filename"][18:])), "rb").read()
    method = "1" if kind == "i" else "0"
    out, args = orc.compress_block(plain, method, b["filename"], "jDC\x01", True)
    assert out == raw
    if kind == "i":
        assert args[:6] == [0, 1, 5, 0, 3, 20]


def test_lz77_decode_of_fixture_iblocks():
    for k in (3, 4, 5):
        b = BLOCKS[k]
        raw = ARC[b["offset"]: b["offset"] + b["size"]]
        plain, meta = orc.decompress_block(raw, 1 << 20)
        assert meta[2] == 2 and len(plain) == b["usize"]
        assert orc.sha1(plain).hex() == b["sha1"]
