"""GPU parity tests added in round 2: the device-resident decode path, hostile LZ77 streams, E8E9 and blocks above
16 MiB (decode-pinned against the real reference), file checksums, long-extent SHA-256, gather without limits.
Run on the MI355X box: python -m pytest tests -m gpu"""
import ctypes as C
import hashlib
import os
import struct
import zlib

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


def exe_like(n, seed):
    a = bytearray(datagen.binary_like(n, seed))
    rng = np.random.default_rng(seed)
    for p in rng.integers(0, n - 8, n // 40):
        a[p] = 0xE8 if p & 2 else 0xE9
        a[p + 4] = 0 if p & 1 else 0xFF
    if n > 3000:
        a[1000:1100] = b"\xe8" * 100
        a[2040:2060] = b"\xe8\x00\xff\xe9" * 5
    return bytes(a)


# ---------------------------------------------------------------------------------------------------
# Decompresser, device-resident (zpq_decompress_blocks_dev)
# ---------------------------------------------------------------------------------------------------
def test_resident_decode_roundtrip_mixed_batch(eng):
    blocks = [datagen.mixed(700000, 1), b"", b"a", datagen.text_like(65536 * 3, 2), datagen.random_bytes(200001, 3),
              datagen.binary_like((1 << 20) + 17, 4), datagen.mixed(5 << 20, 5), bytes(300000)]
    methods = ["14", "0", "14", "0", "14", "x4,1,4,0,2,16", "14", "1"]
    names = ["jDC20240101000000d%010d" % (i + 1) for i in range(len(blocks))]
    framed = eng.compress_blocks(blocks, methods, names, ["jDC\x01"] * len(blocks), True)
    assert all(st == 0 for st, _ in framed)
    res = eng.decompress_blocks_resident([f for _, f in framed], [len(b) + 64 for b in blocks])
    for b, (_, f), r in zip(blocks, framed, res):
        assert r["status"] == 0 and r["data"] == b and r["consumed"] == len(f) and r["sha1"] == orc.sha1(b)
    # the host-buffer entry point is the same path behind a staging copy
    res2 = eng.decompress_blocks([f for _, f in framed], [len(b) + 64 for b in blocks])
    assert [(r["status"], r["data"]) for r in res2] == [(0, b) for b in blocks]


def test_resident_decode_equals_oracle_blocks(eng):
    """Blocks written by the ORACLE (not by this engine) decode to the same bytes."""
    blocks = [datagen.mixed(400000, 11), datagen.text_like(100000, 12), datagen.random_bytes(70000, 13)]
    framed = [orc.compress_block(b, m, "x", "c", True)[0] for b, m in zip(blocks, ["14", "0", "1"])]
    res = eng.decompress_blocks_resident(framed, [len(b) + 64 for b in blocks])
    assert [(r["status"], r["data"]) for r in res] == [(0, b) for b in blocks]


def test_resident_decode_error_statuses(eng):
    b = datagen.mixed(300000, 21)
    (st, f), = eng.compress_blocks([b], ["14"], ["n"], ["c"], True)
    assert st == 0
    bad_sha = bytearray(f); bad_sha[-5] ^= 1
    trunc = f[: len(f) // 2]
    notag = b"\0" * 100
    flip = bytearray(f); flip[len(f) // 2] ^= 0x10                    # damaged LZ77 stream: wrong bytes or a format error, never a crash
    res = eng.decompress_blocks_resident([f, bytes(bad_sha), trunc, notag, f, bytes(flip)], [len(b) + 64] * 4 + [1000, len(b) + 64])
    assert res[0]["status"] == 0 and res[0]["data"] == b
    assert res[1]["status"] == -7 and res[1]["data"] == b             # ZPQ_ERR_CHECKSUM, data still delivered
    assert res[2]["status"] == -6 and res[3]["status"] == -6          # ZPQ_ERR_FORMAT
    assert res[4]["status"] == -4                                     # ZPQ_ERR_CAPACITY
    assert res[5]["status"] in (-7, -6, -4)
    # without verification the checksum is reported, not enforced
    r, = eng.decompress_blocks_resident([bytes(bad_sha)], [len(b) + 64], verify=False)
    assert r["status"] == 0 and r["sha1"] == orc.sha1(b)


def test_resident_decode_failed_pass_block_writes_nothing(eng):
    """A stored (PASS) block that fails AFTER some of its sub-blocks were accepted -- broken trailer, truncated
    sub-block, or more payload than the caller's capacity -- must not copy anything: the outputs of one call sit next
    to each other in one device buffer, so a stray gather would land in the neighbour's bytes."""
    big = datagen.random_bytes(300000, 31)
    (st, f), = eng.compress_blocks([big], ["0"], ["n"], ["c"], True)
    assert st == 0
    cut_marker = bytearray(f); cut_marker[-22] = 0x77                  # 253 -> garbage: "missing 253/254 marker"
    cut_end = f[:-1] + b"\x01"                                         # end-of-block byte replaced: another segment
    trunc = f[: len(f) - 40000]                                        # last sub-block runs past the end
    n = 3
    small = 1000
    # ONE output buffer: [guard | small target | guard] per job, so that an overrun is visible
    from zpaqfranz_amd.engine import UnblockJob
    jobs = (UnblockJob * n)()
    ins = [eng.upload(bytes(b)) for b in (cut_marker, cut_end, trunc)]
    slab = eng.alloc(n * 400000)
    try:
        eng._ck(eng.L.zpq_dev_memset(eng.ctx, slab.ptr, 0xEE, n * 400000))
        for i in range(n):
            jobs[i].in_, jobs[i].n = ins[i].ptr, len((cut_marker, cut_end, trunc)[i])
            jobs[i].out, jobs[i].out_cap = slab.ptr + i * 400000 + 4096, small
        eng.decompress_blocks_dev(jobs, n, True)
        assert all(jobs[i].status < 0 for i in range(n))
        raw = slab.download(n * 400000)
        assert raw == b"\xEE" * (n * 400000), "a failed block copied data"
        # and with room for the payload the broken framing is still refused without a copy
        for i in range(n):
            jobs[i].out_cap = 390000
        eng.decompress_blocks_dev(jobs, n, True)
        assert all(jobs[i].status < 0 for i in range(n))
        assert slab.download(n * 400000) == b"\xEE" * (n * 400000)
    finally:
        for b in ins + [slab]:
            b.free()


def test_resident_decode_fixture_blocks(eng):
    """The c, h and i blocks of the reference's own archive through the device-resident path."""
    arc = open(os.path.join(G, "sha256.zpaq"), "rb").read()
    import json
    meta = [b for b in json.load(open(os.path.join(G, "blocks.json"))) if b["filename"][17] in "chi"]
    framed = [arc[b["offset"]:b["offset"] + b["size"]] for b in meta]
    want = {"h": open(os.path.join(G, "hblock_plain.bin"), "rb").read()}
    res = eng.decompress_blocks_resident(framed, [b["usize"] + 64 for b in meta])
    for b, f, r in zip(meta, framed, res):
        assert r["status"] == 0 and len(r["data"]) == b["usize"] and r["sha1"].hex() == b["sha1"] and r["consumed"] == b["size"]
        if b["filename"][17] == "h":
            assert r["data"] == want["h"]
        if b["filename"][17] == "i":
            assert r["data"] == open(os.path.join(G, "iblock%d.bin" % int(b["filename"][18:])), "rb").read()


def test_many_blocks_one_call(eng):
    """300 small blocks in one call: one gather, one decoder launch, one checksum launch."""
    rng = np.random.default_rng(5)
    blocks = [datagen.mixed(int(rng.integers(1, 90000)), 100 + i) for i in range(300)]
    framed = eng.compress_blocks(blocks, ["14" if i % 3 else "0" for i in range(300)], None, None, True)
    res = eng.decompress_blocks_resident([f for _, f in framed], [len(b) + 64 for b in blocks])
    assert all(r["status"] == 0 and r["data"] == b for r, b in zip(res, blocks))


# ---------------------------------------------------------------------------------------------------
# LZ77 decoder: hostile streams
# ---------------------------------------------------------------------------------------------------
class Bits:
    def __init__(self): self.v = 0; self.n = 0
    def put(self, x, k): self.v |= (x & ((1 << k) - 1)) << self.n; self.n += k
    def bytes(self): return self.v.to_bytes((self.n + 7) // 8, "little")


def _gamma(bs, val):
    ll = val.bit_length() - 1
    for k in range(ll - 1, -1, -1):
        bs.put(1, 1); bs.put((val >> k) & 1, 1)
    bs.put(0, 1)


def test_lz77_decoder_rejects_hostile_lengths(eng):
    # literal run claiming 2^20 bytes with 16 available -> the run is cut at the end of the stream (as the oracle does)
    bs = Bits(); bs.put(0, 2); _gamma(bs, 1 << 20); [bs.put(65 + i, 8) for i in range(16)]
    (st, out), = eng.lz77_decode([bs.bytes()], [1 << 21])
    assert st == 0 and out == bytes(range(65, 81)) == orc.lz77_decode(bs.bytes(), 1 << 21)
    # a literal length code of 26 doublings (no encoder emits more than 13): format error
    bs = Bits(); bs.put(0, 2); _gamma(bs, 1 << 26); [bs.put(65 + i, 8) for i in range(16)]
    (st, out), = eng.lz77_decode([bs.bytes()], [1 << 20])
    assert st == -6
    # match with a length code of 30 doublings: format error
    bs = Bits(); bs.put(0, 2); _gamma(bs, 8); [bs.put(66, 8) for _ in range(8)]
    bs.put(1, 2); bs.put(0, 3)
    for _ in range(30): bs.put(1, 1); bs.put(1, 1)
    bs.put(0, 1); bs.put(0, 2); bs.put(0, 0)
    (st, out), = eng.lz77_decode([bs.bytes() + b"\0" * 16], [1 << 20])
    assert st == -6
    # match longer than the output capacity: capacity error, no write past the buffer (u64 check)
    bs = Bits(); bs.put(0, 2); _gamma(bs, 8); [bs.put(67, 8) for _ in range(8)]
    bs.put(1, 2); bs.put(2, 3); _gamma(bs, (1 << 24) - 1); bs.put(3, 2); bs.put(0, 2)       # off = 4, len ~ 2^26
    (st, out), = eng.lz77_decode([bs.bytes() + b"\0" * 16], [4096])
    assert st == -4
    # offset reaching before the start of the output: format error
    bs = Bits(); bs.put(0, 2); _gamma(bs, 4); [bs.put(68, 8) for _ in range(4)]
    bs.put(2, 2); bs.put(0, 3); _gamma(bs, 1); bs.put(0, 2); bs.put(0x55, 8)               # 8 offset bits -> off >= 256 > 4
    (st, out), = eng.lz77_decode([bs.bytes() + b"\0" * 16], [4096])
    assert st == -6


def test_lz77_decoder_far_match_longer_than_offset(eng):
    """A match whose source lies beyond the 64 KiB LDS ring and whose length exceeds its offset (periodic
    extension through HBM): pieces must only read what is already written."""
    n0 = 70000
    head = datagen.random_bytes(n0, 9)
    bs = Bits()
    pos = 0
    while pos < n0:                                       # literals in runs of 4096
        k = min(4096, n0 - pos)
        bs.put(0, 2); _gamma(bs, k)
        for c in head[pos:pos + k]: bs.put(c, 8)
        pos += k
    off, ln = 69000, 150000                               # source starts 69000 back, length 150000 > off
    lo = off.bit_length() - 1
    bs.put((lo + 8) >> 3, 2); bs.put(lo & 7, 3); _gamma(bs, ln >> 2); bs.put(ln & 3, 2); bs.put(off, lo)
    want = bytearray(head)
    for i in range(ln): want.append(want[len(want) - off])
    (st, out), = eng.lz77_decode([bs.bytes()], [len(want) + 64])
    assert st == 0 and out == bytes(want)
    assert orc.lz77_decode(bs.bytes(), len(want) + 64) == bytes(want)


def test_lz77_decoder_equals_oracle_on_real_streams(eng):
    blocks = [datagen.mixed(3 << 20, 31), datagen.text_like(900000, 32), datagen.random_bytes(300000, 33), bytes(1 << 20), b"ab" * 300000]
    streams = [orc.lz77_encode(b, [4, 1, 5, 0, 3, 24]) for b in blocks]
    res = eng.lz77_decode(streams, [len(b) + 64 for b in blocks])
    assert [(s, o) for s, o in res] == [(0, b) for b in blocks]
    # truncated streams: whatever the oracle's decoder delivers
    for cut in (1, 7, 100, 5000):
        s = streams[0][:-cut]
        (st, out), = eng.lz77_decode([s], [len(blocks[0]) + 64])
        assert st == 0 and out == orc.lz77_decode(s, len(blocks[0]) + 64)


def test_lz77_encoder_segment_size_never_changes_the_stream(eng, monkeypatch):
    """The speculation segment is a per-call choice (1 MiB by default, growing to one segment per block when many
    blocks share the HBM budget); the code stream must not depend on it, nor on how the call is cut into batches."""
    blocks = [datagen.mixed((5 << 20) + 777, 41), datagen.text_like(3 << 20, 42), datagen.binary_like(700000, 43)]
    args = [4, 1, 5, 0, 3, 24]
    want = [orc.lz77_encode(b, args) for b in blocks]
    for seg in ("262144", "1048576", "4194304", "1073741824"):
        monkeypatch.setenv("ZPQ_LZ_SEG", seg)
        assert eng.lz77_encode(blocks, [args] * 3) == want
    monkeypatch.setenv("ZPQ_LZ_BUDGET_MB", "300")                 # every block its own batch
    assert eng.lz77_encode(blocks, [args] * 3) == want
    monkeypatch.delenv("ZPQ_LZ_SEG")
    assert eng.lz77_encode(blocks, [args] * 3) == want            # automatic segment choice under a tight budget
    monkeypatch.delenv("ZPQ_LZ_BUDGET_MB")
    # direct mode (one wave per block parses AND emits: what a call with hundreds of blocks selects by itself)
    monkeypatch.setenv("ZPQ_LZ_DIRECT", "1")
    more = blocks + [b"", b"a", b"abcd" * 3, bytes(70000), datagen.random_bytes(300000, 44), b"ab" * 150000, datagen.text_like(4097, 45)]
    assert eng.lz77_encode(more, [args] * len(more)) == want + [orc.lz77_encode(b, args) for b in more[3:]]
    for a2 in ([5, 1, 5, 0, 3, 25], [4, 1, 4, 0, 1, 18], [2, 1, 6, 0, 0, 20], [4, 1, 8, 0, 2, 22]):      # rb = 1; other bucket widths and match lengths
        assert eng.lz77_encode(blocks[1:], [a2] * 2) == [orc.lz77_encode(b, a2) for b in blocks[1:]]
    monkeypatch.setenv("ZPQ_LZ_BUDGET_MB", "100")                 # one block per batch
    assert eng.lz77_encode(blocks, [args] * 3) == want
    monkeypatch.delenv("ZPQ_LZ_BUDGET_MB")
    monkeypatch.delenv("ZPQ_LZ_DIRECT")


# ---------------------------------------------------------------------------------------------------
# E8E9 and blocks above 16 MiB (rb > 0): decode-pinned against the real reference
# ---------------------------------------------------------------------------------------------------
def test_e8e9_inverse_equals_oracle(eng, monkeypatch):
    for n, seed in ((0, 1), (4, 2), (5, 3), (6, 4), (1023, 5), (1024, 6), (1030, 7), (300000, 8), ((1 << 20) + 3, 9)):
        data = exe_like(n, seed) if n > 8 else bytes(range(0xE6, 0xE6 + n))
        t = orc.e8e9(data)
        for wu in ("64", "0", "3"):
            monkeypatch.setenv("ZPQ_E8_WARMUP", wu)                       # 0: every segment assumes a clean start, the fixer repairs it
            d_in = eng.upload(t); d_out = eng.alloc(max(1, n))
            L = eng.L
            L.zpq_e8e9_inverse_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
            assert L.zpq_e8e9_inverse_dev(eng.ctx, d_in.ptr, d_out.ptr, n) == 0
            got = d_out.download(n)
            d_in.free(); d_out.free()
            assert got == data, (n, wu)
    monkeypatch.delenv("ZPQ_E8_WARMUP")


@needs_ref
@pytest.mark.parametrize("method,n", [("x4,5,5,0,3,24", 700000), ("x5,1,5,0,3,25", 900000), ("x6,5,4,0,2,22", 500000), ("x5,5,6,0,3,25", (17 << 20) + 12345)])
def test_compress_block_e8e9_and_big_blocks(eng, method, n):
    data = exe_like(n, 77) if ",5," in method[:5] else datagen.mixed(n, 78)
    (st, framed), = eng.compress_blocks([data], [method], ["blk"], ["c"], True)
    assert st == 0
    # (1) the REAL reference Decompresser restores the input and accepts the checksum
    r = orc.ref_decompress_block(framed, n + 64)
    assert r["data"] == data and r["sha1_ok"] == 1 and r["consumed"] == len(framed)
    # (2) the code stream inside is the real LZBuffer's for these arguments
    from zpaqfranz_amd import engine
    _, args = engine.make_config(method)
    lz = orc.ref_lzbuffer(data, args)
    payload, p = b"", framed.index(b"\0\0", framed.index(b"blk")) + 2
    p = framed.index(b"blk") + 4
    p = framed.index(b"\0", p) + 2                                   # past comment NUL and the reserved byte
    while True:
        k = struct.unpack(">I", framed[p:p + 4])[0]; p += 4
        if not k: break
        payload += framed[p:p + k]; p += k
    psize = payload[1] | payload[2] << 8
    assert payload[0] == 1 and payload[3 + psize:] == lz
    # (3) both decode paths of this engine restore it natively
    for res in (eng.decompress_blocks_resident([framed], [n + 64]), eng.decompress_blocks([framed], [n + 64])):
        assert res[0]["status"] == 0 and res[0]["data"] == data


# ---------------------------------------------------------------------------------------------------
# file checksums (section 8f-2)
# ---------------------------------------------------------------------------------------------------
def test_file_checksums_fixture_attributes(eng):
    from test_checksum_cpu import golden_files
    files = golden_files()
    crc, xx, b3 = eng.file_checksums([f for f, _, _ in files])
    assert ["%08X" % c for c in crc] == [c for _, _, c in files]
    assert ["%016X" % x for x in xx] == [x for _, x, _ in files]
    assert b3 == [orc.blake3(f) for f, _, _ in files]


def test_file_checksums_ragged(eng):
    xxhash = pytest.importorskip("xxhash")
    sizes = [0, 1, 3, 4, 7, 8, 31, 32, 33, 63, 64, 65, 255, 256, 1023, 1024, 1025, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193,
             65535, 65536, 65537, 100003, 64 * 4096, 64 * 4096 + 1, 65 * 4096 + 3, (1 << 20) - 1, 1 << 20, (5 << 20) + 12345, 0, 2]
    files = [datagen.random_bytes(n, 1000 + i) for i, n in enumerate(sizes)]
    crc, xx, b3 = eng.file_checksums(files)
    assert crc == [zlib.crc32(f) for f in files]
    assert xx == [xxhash.xxh64(f).intdigest() for f in files]
    assert b3 == [orc.blake3(f) for f in files]
    pat = bytes(i % 251 for i in range(102400))
    _, _, kb = eng.file_checksums([pat[:0], pat[:1], pat[:1024], pat[:1025], pat[:2048], pat], crc32=False, xxh64=False)
    assert [k.hex()[:16] for k in kb] == ["af1349b9f5f9a1a6", "2d3adedff11b61f1", "42214739f095a406", "d00278ae47eb27b3", "e776b6028c7cd22a", "bc3e3d41a1146b06"]


def test_file_checksums_many_files(eng):
    rng = np.random.default_rng(4)
    files = [datagen.random_bytes(int(rng.integers(0, 20000)), 5000 + i) for i in range(3000)]
    crc, xx, b3 = eng.file_checksums(files)
    assert crc == [zlib.crc32(f) for f in files]
    assert xx == [orc.xxh64(f) for f in files]
    assert b3[::37] == [orc.blake3(f) for f in files[::37]]


# ---------------------------------------------------------------------------------------------------
# SHA-256 of long extents (one wave per file), gather without limits, digest compare
# ---------------------------------------------------------------------------------------------------
def test_sha256_long_and_short_extents_together(eng, monkeypatch):
    sizes = [(1 << 20), (1 << 20) - 1, (1 << 20) + 1, (3 << 20) + 77, 64 * 64 * 7, 5, 0, (2 << 20) + 64, 4096, (1 << 20) + 63, (1 << 20) + 64 * 64]
    files = [datagen.random_bytes(n, 300 + i) for i, n in enumerate(sizes)]
    assert eng.sha256_many(files) == [hashlib.sha256(f).digest() for f in files]
    monkeypatch.setenv("ZPQ_SHA256_CHAIN_MIN", "1")              # everything on the wave-wide kernel, tails of every length
    small = [datagen.random_bytes(n, 400 + n) for n in list(range(0, 200)) + [4096, 4097, 8191]]
    assert eng.sha256_many(small) == [hashlib.sha256(f).digest() for f in small]
    monkeypatch.delenv("ZPQ_SHA256_CHAIN_MIN")


@pytest.mark.parametrize("lanes", [4, 8, 16, 32])
def test_sha256_several_chains_per_wave(eng, monkeypatch, lanes):
    """sha256_group_kernel: 64 / lanes chains per wave (what an extract's thousands of restored files go through): more
    chains than groups (the queue), unequal lengths inside a wave, every tail length, batches that end inside a group's
    `lanes` blocks, plus short extents on the lane-wise kernel beside them."""
    monkeypatch.setenv("ZPQ_SHA256_GROUP", str(lanes))
    monkeypatch.setenv("ZPQ_SHA256_CHAIN_MIN", "4096")
    sizes = [4096 + 64 * k + (k * 37) % 64 for k in range(0, 70)] + [64 * lanes * 5, 64 * lanes * 5 + 1, 64 * lanes * 5 - 1, 200000, 4096, 77777,
                                                                     (1 << 18) + 13, 5, 0, 4095, 100, 64, 63, 4160]
    files = [datagen.random_bytes(n, 900 + i) for i, n in enumerate(sizes)]
    assert eng.sha256_many(files) == [hashlib.sha256(f).digest() for f in files]
    # one long chain alone in its wave, and exactly one group's worth of chains
    files = [datagen.random_bytes(300000 + 7 * i, 990 + i) for i in range(64 // lanes)]
    assert eng.sha256_many(files[:1]) == [hashlib.sha256(files[0]).digest()]
    assert eng.sha256_many(files) == [hashlib.sha256(f).digest() for f in files]


def test_gather_long_extents_and_many_extents(eng):
    rng = np.random.default_rng(8)
    src = datagen.random_bytes(12 << 20, 55)
    d_src = eng.upload(src)
    for lens in ([3 << 20, 1, (1 << 20) + 5, 0, 70000, (2 << 20) - 1], rng.integers(0, 300, 20000).tolist()):
        n = len(lens)
        so = rng.integers(0, (12 << 20) - max(lens) - 1, n).astype(np.uint64)
        do = np.concatenate(([0], np.cumsum(np.array(lens, dtype=np.uint64) + 3)))[:-1].astype(np.uint64)   # unaligned destinations
        total = int(do[-1]) + lens[-1] + 64
        d_so, d_sl, d_do = eng.upload(so.tobytes()), eng.upload(np.array(lens, dtype=np.uint32).tobytes()), eng.upload(do.tobytes())
        d_dst = eng.alloc(total)
        eng._ck(eng.L.zpq_dev_memset(eng.ctx, d_dst.ptr, 0xAA, total))
        eng.gather_dev(d_src.ptr, d_so.ptr, d_sl.ptr, d_do.ptr, n, d_dst.ptr)
        eng.sync()
        got = d_dst.download(total)
        want = bytearray(b"\xaa" * total)
        for s, l, d in zip(so.tolist(), lens, do.tolist()):
            want[d:d + l] = src[s:s + l]
        assert got == bytes(want)
        for b in (d_so, d_sl, d_do, d_dst): b.free()
    d_src.free()


def test_digest_compare(eng):
    a = datagen.random_bytes(20 * 5000, 1)
    b = bytearray(a); b[20 * 1234 + 7] ^= 1; b[20 * 4000] ^= 0x80
    da, db = eng.upload(a), eng.upload(bytes(b))
    assert eng.digest_compare_dev(da.ptr, da.ptr, 5000, 20) == (0, 0)
    assert eng.digest_compare_dev(da.ptr, db.ptr, 5000, 20) == (2, 1234)
    da.free(); db.free()


# ---------------------------------------------------------------------------------------------------
# libzpaq shim: Compressor (streaming writer), multi-segment stored blocks, N threads decoding at once
# ---------------------------------------------------------------------------------------------------
@needs_ref
def test_shim_compressor_and_parallel_extract(tmp_path):
    import subprocess
    import cmconfigs
    from zpaqfranz_amd import build, engine
    build.build(verbose=False)
    drv = build.build_shim_driver(str(tmp_path / "shim_driver"))
    data = datagen.mixed(300000, 91)
    (tmp_path / "in.bin").write_bytes(data)
    # (a) stored model, three segments in one block: the real reference decoder reads all of them, checksums match
    (tmp_path / "store.cfg").write_text("comp 0 0 0 0 0 hcomp end\n")
    r = subprocess.run([drv, "--compressor", str(tmp_path / "in.bin"), str(tmp_path / "store.cfg"), "3", str(tmp_path / "s.zpaq")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    arc = (tmp_path / "s.zpaq").read_bytes()
    ref = orc.ref_decompress_block(arc, len(data) + 64)
    assert ref["data"] == data and ref["segments"] == 3 and ref["sha1_ok"] == 1 and ref["consumed"] == len(arc)
    # ... and so does this shim's own decompress(), from 6 threads at once (decode batcher, SHA-1 batcher)
    r = subprocess.run([drv, "--parallel-extract", str(tmp_path / "s.zpaq"), str(tmp_path / "in.bin"), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "parallel extract ok" in r.stdout, r.stderr
    # (b) a context model from source: the coded bytes are the reference Encoder's over 0 (PASS) + data
    small = data[:20000]
    (tmp_path / "small.bin").write_bytes(small)
    (tmp_path / "mid.cfg").write_text(cmconfigs.MID)
    r = subprocess.run([drv, "--compressor", str(tmp_path / "small.bin"), str(tmp_path / "mid.cfg"), "1", str(tmp_path / "m.zpaq")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    arc = (tmp_path / "m.zpaq").read_bytes()
    hdr, _ = engine.compile_config(cmconfigs.MID)
    coded = orc.ref_cm_encode(hdr, b"\0" + small)
    want = bytes.fromhex("376b5374a03183d38cb228b0d3") + b"zPQ\x01\x01" + hdr + b"\x01seg0\0c\0\0" + coded + b"\xfd" + orc.sha1(small) + b"\xff"
    assert arc == want
    assert orc.ref_decompress(arc, len(small) + 64) == small
    # (c) blocks written by compressBlock from 4 threads, decoded by 6 threads at once
    r = subprocess.run([drv, str(tmp_path / "in.bin"), "14", "4", "70000", str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([drv, "--parallel-extract", str(tmp_path / "out.zpaq"), str(tmp_path / "in.bin"), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "parallel extract ok" in r.stdout, r.stderr


# ---------------------------------------------------------------------------------------------------
# journaling engine: several contexts (one per GPU; here all on GPU 0), device-resident extract, hostile archives
# ---------------------------------------------------------------------------------------------------
def test_jidac_add_multi_is_identical_to_single_context(eng):
    from zpaqfranz_amd import Engine, engine as E
    shared = datagen.mixed(2 << 20, 61)
    files = [("a/one", datagen.text_like(1500000, 62)), ("a/two", shared + datagen.binary_like(300000, 63)), ("b/three", shared),
             ("c/empty", b""), ("d/four", datagen.mixed(3 << 20, 64)), ("e/five", datagen.random_bytes(700000, 65)), ("f/six", shared[: 1 << 20])]
    one, st1 = E.jidac_add(eng, b"", files, 20240101120000)
    extra = [Engine(0), Engine(0)]
    try:
        for engs in ([eng, extra[0]], [eng, extra[0], extra[1]]):
            many, stn = E.jidac_add(engs, b"", files, 20240101120000)
            assert many == one and stn == st1
        # a second version on top, sharded: dedup against the index of version 1
        files2 = files + [("g/new", datagen.text_like(400000, 66) + shared[:500000])]
        v2a, _ = E.jidac_add(eng, one, files2, 20240202120000)
        v2b, _ = E.jidac_add([eng, extra[0]], one, files2, 20240202120000)
        assert v2a == v2b
        assert E.jidac_extract(eng, one + v2a) == {n: d for n, d in files2}
    finally:
        for e in extra:
            e.close()


def test_jidac_add_dev_equals_the_host_pointer_add(eng):
    """zpqj_add_dev (files already in HBM, back to back in name order) returns the archive zpqj_add / zpqj_add_opts return for
    the same files from host pointers -- byte for byte: first version, a second version on top of it, with the checksum
    attributes, with the twin fold off -- and the real reference decoder walks it; names out of order are refused."""
    from zpaqfranz_amd import engine as E
    shared = datagen.mixed(2 << 20, 71)
    files = [("a/one", datagen.text_like(1200000, 72)), ("a/two", shared + datagen.binary_like(300000, 73)), ("b/three", shared),
             ("b/three.copy", shared), ("c/empty", b""), ("d/four", datagen.mixed(2 << 20, 74)), ("e/five", datagen.random_bytes(500000, 75))]
    assert [n for n, _ in files] == sorted(n for n, _ in files)

    def resident(fs):
        off = [0]
        for _, d in fs:
            off.append(off[-1] + len(d))
        return eng.upload(b"".join(d for _, d in fs)), E.DevFiles([n for n, _ in fs], off, version_date=20240101120000)
    buf, df = resident(files)
    try:
        want, st_w = E.jidac_add(eng, b"", files, 20240101120000)
        got, st_g = E.jidac_add_dev(eng, b"", buf.ptr, df, 20240101120000)
        assert got == want and st_g == st_w
        plain, st_p = E.jidac_add_dev(eng, b"", buf.ptr, df, 20240101120000, twins=False)       # every byte hashed: same archive
        assert plain == want and st_p == st_w
        want_c, _ = E.jidac_add(eng, b"", files, 20240101120000, checksums=True, hint=True)
        got_c, _ = E.jidac_add_dev(eng, b"", buf.ptr, df, 20240101120000, checksums=True, hint=True)
        assert got_c == want_c and got_c != want
        assert E.jidac_extract(eng, got) == {n: d for n, d in files}
        # a second version on top
        files2 = list(files) + [("g/new", datagen.text_like(300000, 76) + shared[:400000])]
        files2[0] = ("a/one", files[0][1][:300000] + b"EDIT" + files[0][1][300000:])
        buf2, df2 = resident(files2)
        try:
            df2.dates = (C.c_int64 * len(files2))(*([20240202120000] * len(files2)))
            v2w, _ = E.jidac_add(eng, want, files2, 20240202120000)
            v2g, _ = E.jidac_add_dev(eng, want, buf2.ptr, df2, 20240202120000)
            assert v2g == v2w
            assert E.jidac_extract(eng, want + v2g) == {n: d for n, d in files2}
        finally:
            buf2.free()
        # names that do not ascend: refused (the buffer order would not be the order the files are walked in)
        bad = E.DevFiles(["b", "a"], [0, 10, 20], version_date=1)
        with pytest.raises(E.ZpqError):
            E.jidac_add_dev(eng, b"", buf.ptr, bad, 20240101120000)
        # ... the same name twice, offsets that go backwards, a base the 16-byte reads cannot use: refused, each of them
        for names_, off_, base_ in ((["a", "a"], [0, 10, 20], buf.ptr), (["a", "b"], [0, 20, 10], buf.ptr), (["a", "b"], [0, 10, 20], buf.ptr + 4)):
            with pytest.raises(E.ZpqError):
                E.jidac_add_dev(eng, b"", base_, E.DevFiles(names_, off_, version_date=1), 20240101120000)
        if orc.have_ref():
            total = sum(len(d) for _, d in files)
            blocks, off = [], 0
            while off < len(got):
                b = orc.ref_decompress_block(got[off:], total + 65536)
                assert b["sha1_ok"] == 1
                blocks.append(b); off += b["consumed"]
            assert "".join(chr(b["filename"][17]) for b in blocks).startswith("cd")
            # bench.py's own walker of such archives (verification outside its timed region) sees the same blocks
            import sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            mine = bench.split_archive(got)
            assert [(m[0], m[3] - m[2]) for m in mine] == [(b["filename"], b["consumed"]) for b in blocks]
            assert all(m[4] == b"\0" + b["data"] for m, b in zip(mine, blocks) if chr(m[0][17]) in "ch")      # (method 0: the PASS byte, then the bytes)
    finally:
        buf.free()


def test_jidac_rejects_hostile_archives(eng):
    from zpaqfranz_amd import engine as E, ZpqError
    files = [("x", datagen.mixed(400000, 71)), ("y", datagen.text_like(100000, 72))]
    arc, _ = E.jidac_add(eng, b"", files, 20240101120000)
    assert E.jidac_extract(eng, arc) == dict(files)
    # a comment claiming an absurd size, a truncated tail, a damaged d block: errors, never a crash or an exception across the ABI
    bad = bytearray(arc)
    at = bad.index(b" jDC\x01")
    bad2 = bytes(bad[:at - 1]) + b"99999999999999999999" + bytes(bad[at:])
    for broken in (bad2, arc[: len(arc) - 7], arc[:40], arc[:200] + b"\xff" * 50 + arc[250:]):
        with pytest.raises(ZpqError):
            E.jidac_extract(eng, broken)


@pytest.mark.gpu
@pytest.mark.parametrize("staged", ["1", "0"])
def test_sha1_extents_staged_and_direct_forms_agree_with_hashlib(staged):
    """The fragment SHA-1 pass in both forms (wave-fetched through LDS = default for > 4096 extents, lane loads = fallback)
    on fragment-shaped extents (exponential lengths 4 KiB..508 KiB, back to back) and on equal extents scattered over the
    buffer; 52 digests of each (longest, shortest, first, last, random) against hashlib.  The form is chosen once per
    process, hence the subprocess."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZPQ_SHA1_STAGED=staged)
    for kind in ("frag", "scattered"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "sha1_extents_probe.py"), kind, "1.5"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "digests ok: True" in r.stdout and ("staged=%s" % staged) in r.stdout, r.stdout


@pytest.mark.parametrize("waves", ["1", "2"])
def test_sha1_extents_staged_form_without_torch(waves):
    """The wave-fetched fragment SHA-1 (> 4096 extents): 5000 extents of 0..700 bytes plus a few long ones, ragged starts, every
    digest against hashlib, with four and with eight waves per compute unit (ZPQ_SHA_WAVES).  Plain C ABI, no torch: runs on the
    emulated engine too.  (The grid is read per process, hence the subprocess.)"""
    import subprocess, sys
    code = r'''
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.environ["ZPQ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["ZPQ_ROOT"], "tests"))
import numpy as np
from zpaqfranz_amd import Engine
rng = np.random.default_rng(9)
lens = rng.integers(0, 700, size=5000).astype(np.uint32)
lens[[7, 1234, 4999]] = [70001, 128, 4097]; lens[[11, 12, 13]] = [64, 63, 65]
gaps = rng.integers(0, 9, size=5000)
off = np.cumsum(np.concatenate(([3], (lens[:-1] + gaps[:-1]).astype(np.int64)))).astype(np.uint64)
total = int(off[-1] + lens[-1])
data = rng.integers(0, 256, size=total, dtype=np.uint8).tobytes()
eng = Engine(0)
d, do, dl, dg = eng.upload(data), eng.upload(off.tobytes()), eng.upload(lens.tobytes()), eng.alloc(5000 * 20 + 64)
eng.sha1_extents_dev(d.ptr, do.ptr, dl.ptr, 5000, dg.ptr)
eng.sync()
got = dg.download(5000 * 20)
bad = [i for i in range(5000) if got[20 * i:20 * i + 20] != hashlib.sha1(data[int(off[i]):int(off[i]) + int(lens[i])]).digest()]
print("bad", bad[:5], len(bad))
sys.exit(1 if bad else 0)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZPQ_SHA_WAVES=waves, ZPQ_SHA1_STAGED="1", ZPQ_ROOT=root),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("level", [1, 2])
def test_shim_compressor_startblock_level(tmp_path, level):
    """Compressor::startBlock(int level) (ZSFX/libzpaq.h:1346): the block carries libzpaq's built-in model `level`, the coded
    bytes are the reference Encoder's for that model, the real reference decoder restores the input."""
    import subprocess
    import ctypes as C
    from zpaqfranz_amd import build, engine
    build.build(verbose=False)
    drv = build.build_shim_driver(str(tmp_path / "shim_driver"))
    small = datagen.mixed(20000, 92 + level)
    (tmp_path / "small.bin").write_bytes(small)
    r = subprocess.run([drv, "--compressor", str(tmp_path / "small.bin"), "level:%d" % level, "1", str(tmp_path / "l.zpaq")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    arc = (tmp_path / "l.zpaq").read_bytes()
    L = engine.load()
    L.zpq_builtin_model.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(512)
    n = C.c_size_t(0)
    assert L.zpq_builtin_model(level, buf, 512, C.byref(n)) == 0
    hdr = buf.raw[: n.value]
    coded = orc.ref_cm_encode(hdr, b"\0" + small)
    want = bytes.fromhex("376b5374a03183d38cb228b0d3") + b"zPQ\x01\x01" + hdr + b"\x01seg0\0c\0\0" + coded + b"\xfd" + orc.sha1(small) + b"\xff"
    assert arc == want
    assert orc.ref_decompress(arc, len(small) + 64) == small
