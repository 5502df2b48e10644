"""The checksum oracle (oracle/checksum_oracle.cpp) against what pins it: the XXHASH64 / CRC-32 attributes the
reference wrote into the i blocks of its golden archive (AUTOTEST/sha256.zpaq, 256 files), independent
implementations shipped with the image (zlib.crc32, the xxhash module) and the published known answers."""
import lzma
import os
import struct
import zlib

import numpy as np
import pytest

import orc

G = orc.GOLDEN


def golden_files():
    """[(data, xxh64 hex, crc32 hex)] of the 256 files of the fixture, from the i blocks' attributes."""
    dplain = lzma.decompress(open(os.path.join(G, "dblock_plain.xz"), "rb").read())
    h = open(os.path.join(G, "hblock_plain.bin"), "rb").read()
    lens = [struct.unpack("<I", h[24 + 24 * i: 28 + 24 * i])[0] for i in range(388)]
    offs = np.concatenate(([0], np.cumsum(lens))).tolist()
    out = []
    for k in (1, 2, 3):
        ib = open(os.path.join(G, "iblock%d.bin" % k), "rb").read()
        p = 0
        while p < len(ib):
            date = struct.unpack("<q", ib[p:p + 8])[0]; p += 8
            e = ib.index(b"\0", p); p = e + 1
            if date:
                na = struct.unpack("<I", ib[p:p + 4])[0]; p += 4
                attr = ib[p:p + na]; p += na
                ni = struct.unpack("<I", ib[p:p + 4])[0]; p += 4
                ptr = struct.unpack("<%dI" % ni, ib[p:p + 4 * ni]); p += 4 * ni
                out.append((b"".join(dplain[offs[q - 1]:offs[q]] for q in ptr), attr[16:32].decode(), attr[49:57].decode()))
    return out


def test_oracle_reproduces_the_fixture_attributes():
    files = golden_files()
    assert len(files) == 256
    for data, xx, crc in files:
        assert "%016X" % orc.xxh64(data) == xx
        assert "%08X" % orc.crc32(data) == crc


def test_oracle_equals_independent_implementations():
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(3)
    for n in [0, 1, 3, 4, 7, 8, 15, 31, 32, 33, 63, 64, 65, 1000, 4095, 4096, 4097, 100003]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert orc.crc32(b) == zlib.crc32(b)
        assert orc.xxh64(b) == xxhash.xxh64(b).intdigest()


def test_published_known_answers():
    assert orc.crc32(b"123456789") == 0xCBF43926
    assert orc.xxh64(b"") == 0xEF46DB3751D8E999
    pat = lambda n: bytes(i % 251 for i in range(n))           # the input pattern of BLAKE3's test_vectors.json
    kat = {0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
           1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
           1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
           1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
           2048: "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a"}
    for n, want in kat.items():
        assert orc.blake3(pat(n)).hex() == want
    assert orc.blake3(b"abc").hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"


def test_blake3_level_fold_equals_the_recursive_tree():
    """The GPU folds chaining values level by level (pairs, the odd one carried up); the specification defines the
    tree recursively (left subtree = largest power of two).  Same tree: modelled here with the oracle's pieces."""
    import ctypes as C
    L = orc._L

    def chunk_cv(b, idx, root):
        out = (C.c_uint32 * 8)(); L.orc_blake3_chunk_cv(orc._buf(b), C.c_long(len(b)), C.c_uint64(idx), int(root), out); return list(out)

    def parent(l, r, root):
        out = (C.c_uint32 * 8)(); L.orc_blake3_parent((C.c_uint32 * 8)(*l), (C.c_uint32 * 8)(*r), int(root), out); return list(out)

    rng = np.random.default_rng(1)
    for n in [0, 1, 1024, 1025, 3072, 3073, 5000, 65536, 65537, 129 * 1024, 130 * 1024 + 1, 300 * 1024 + 77]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        nch = max(1, (n + 1023) // 1024)
        cvs = [chunk_cv(d[i * 1024:(i + 1) * 1024], i, nch == 1) for i in range(nch)]
        cnt = nch
        while cnt > 1:
            parents, nxt = cnt >> 1, (cnt + 1) >> 1
            cvs = [parent(cvs[2 * j], cvs[2 * j + 1], nxt == 1) if j < parents else cvs[2 * j] for j in range(nxt)]
            cnt = nxt
        assert b"".join(int(w).to_bytes(4, "little") for w in cvs[0]) == orc.blake3(d), n
