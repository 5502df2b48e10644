"""zpaqfranz `t` on the device (zpqj_verify): blocks, fragments and the per-file XXHASH64 / CRC-32 the i blocks carry.
The attribute layout is the one the reference wrote into AUTOTEST/sha256.zpaq; zpqj_add_opts writes the
same shape, with checksums computed by zpq_file_checksums_dev in the pass that fragments the files."""
import struct
import zlib

import pytest

import datagen
import orc

pytestmark = pytest.mark.gpu
TAG = bytes.fromhex("376b5374a03183d38cb228b0d3")


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def files():
    shared = datagen.mixed(1 << 20, 91)
    return [("v/a", datagen.text_like(700000, 92)), ("v/b", shared + datagen.binary_like(200000, 93)), ("v/c", shared), ("v/empty", b""),
            ("v/d", datagen.random_bytes(300001, 94))]


def last_block(arc):
    at = arc.rfind(TAG)
    return at, arc[at:]


def test_verify_plain_archive_has_nothing_stored_to_compare(eng):
    from zpaqfranz_amd import engine as E
    arc, _ = E.jidac_add(eng, b"", files(), 20240101120000)
    rc, st = E.jidac_verify(eng, arc)
    assert rc == 0 and st["files"] == 5 and st["files_with_checksums"] == 0 and st["bytes"] == sum(len(d) for _, d in files())


def test_add_with_checksums_then_verify(eng):
    from zpaqfranz_amd import Engine, engine as E
    fs = files()
    arc, st_add = E.jidac_add(eng, b"", fs, 20240101120000, checksums=True)
    rc, st = E.jidac_verify(eng, arc)
    assert rc == 0 and st["files"] == 5 and st["files_with_checksums"] == 5 and st["xxh64_mismatches"] == 0 and st["crc32_mismatches"] == 0
    assert st["fragments"] == st_add["new_fragments"] and st["d_blocks"] == st_add["d_blocks"]
    assert E.jidac_extract(eng, arc) == dict(fs)
    # the attributes hold what zlib says, in the fixture's layout
    at, blk = last_block(arc)
    ib = eng.decompress_blocks([blk], [1 << 20])[0]["data"]
    p = 0
    seen = {}
    while p < len(ib):
        p += 8
        e = ib.index(b"\0", p); name = ib[p:e].decode(); p = e + 1
        na = struct.unpack("<I", ib[p:p + 4])[0]; p += 4
        attr = ib[p:p + na]; p += na
        ni = struct.unpack("<I", ib[p:p + 4])[0]; p += 4 + 4 * ni
        assert na == 58
        seen[name] = attr[49:57].decode()
    assert seen == {n: "%08X" % zlib.crc32(d) for n, d in fs}
    # two contexts give the same archive
    other = Engine(0)
    try:
        arc2, _ = E.jidac_add([eng, other], b"", fs, 20240101120000, checksums=True)
        assert arc2 == arc
    finally:
        other.close()


@pytest.mark.skipif(not orc.have_ref(), reason="needs oracle/_ref to read the block name back")
def test_verify_reports_a_wrong_stored_checksum(eng):
    """The i block is rewritten with one hex digit of a stored CRC-32 changed: every block and fragment still verifies,
    the file checksum does not."""
    from zpaqfranz_amd import engine as E
    arc, _ = E.jidac_add(eng, b"", files(), 20240101120000, checksums=True)
    at, blk = last_block(arc)
    info = orc.ref_decompress_block(blk, 1 << 20)
    ib = bytearray(info["data"])
    i = ib.index(b"v/a\0") + 4 + 4 + 49
    ib[i] = ord("1") if ib[i] != ord("1") else ord("2")
    (st, fb), = eng.compress_blocks([bytes(ib)], ["1"], [info["filename"].decode()], [info["comment"].split(b" ", 1)[1].decode("latin1")], True)
    assert st == 0
    rc, stv = E.jidac_verify(eng, arc[:at] + fb)
    assert rc == -7 and stv["crc32_mismatches"] == 1 and stv["xxh64_mismatches"] == 0 and stv["files_with_checksums"] == 5


def test_verify_reports_a_damaged_block(eng):
    from zpaqfranz_amd import engine as E
    arc, _ = E.jidac_add(eng, b"", files(), 20240101120000, checksums=True)
    bad = bytearray(arc)
    first_d = arc.index(TAG, arc.index(TAG) + 13)          # the d block follows the c block
    bad[first_d + 2000] ^= 0x10
    rc, _ = E.jidac_verify(eng, bytes(bad))
    assert rc != 0


@pytest.mark.parametrize("twins", [False, True])
def test_extract_with_archive_and_files_resident_in_hbm(eng, twins):
    """zpqj_extract_dev (Jidac::extract, ZSFX/zsfx.cpp:2018-2281, as ONE call with the archive in HBM and the files left there):
    the plan call returns the index, the real call the same index, the files byte for byte and every file's SHA-256 -- for an
    archive of two versions (the second one's d blocks lie behind the first one's index: the c-block jump, :1432-1461)."""
    import hashlib
    from zpaqfranz_amd import engine as E
    fs = files()
    arc1, _ = E.jidac_add(eng, b"", fs, 20240101120000)
    more = [("v/a", datagen.text_like(300000, 95)), ("v/e", fs[2][1] + b"tail"), ("w/twin1", fs[4][1]), ("w/twin2", fs[4][1])]
    arc2, _ = E.jidac_add(eng, arc1, more, 20240102120000)
    arc = arc1 + arc2
    want = dict(fs); want.update(dict(more))
    d_arc = eng.upload(arc)
    names, off, st = E.jidac_extract_dev(eng, d_arc.ptr, len(arc))
    assert names == sorted(want) and off[0] == 0 and [off[i + 1] - off[i] for i in range(len(names))] == [len(want[n]) for n in names]
    assert st["files"] == len(want) and st["bytes"] == off[-1]
    d_out = eng.alloc(off[-1]); d_sha = eng.alloc(32 * len(names))
    names2, off2, st2 = E.jidac_extract_dev(eng, d_arc.ptr, len(arc), d_out.ptr, off[-1] + 64, d_sha.ptr, len(names), twins=twins)
    assert (names2, off2) == (names, off) and st2["fragments"] >= st2["d_blocks"] >= 2
    blob = d_out.download(off[-1])
    assert all(blob[off[i]:off[i + 1]] == want[n] for i, n in enumerate(names))
    sha = d_sha.download(32 * len(names))
    assert [sha[32 * i:32 * i + 32] for i in range(len(names))] == [hashlib.sha256(want[n]).digest() for n in names]
    # a buffer that is too small is refused, and so is a damaged fragment
    with pytest.raises(E.ZpqError):
        E.jidac_extract_dev(eng, d_arc.ptr, len(arc), d_out.ptr, off[-1], d_sha.ptr, len(names))
    bad = bytearray(arc); at = arc.index(TAG, 50) + 400; bad[at] ^= 1
    d_bad = eng.upload(bytes(bad))
    with pytest.raises(E.ZpqError):
        E.jidac_extract_dev(eng, d_bad.ptr, len(arc), d_out.ptr, off[-1] + 64, d_sha.ptr, len(names))
    for d in (d_arc, d_out, d_sha, d_bad):
        d.free()


def test_pool_trim_gives_idle_blocks_back(eng):
    """zpq_dev_alloc_pooled keeps freed blocks in the context; zpq_pool_trim (what a failed allocation of ANY context of the device
    now does before it reports ZPQ_ERR_NOMEM: ADVICE round 5) hands the idle ones back to the driver and leaves blocks in use alone."""
    import ctypes as C
    from zpaqfranz_amd import Engine
    L = eng.L
    L.zpq_pool_trim.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.zpq_dev_alloc_pooled.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.zpq_dev_free_pooled.argtypes = [C.c_void_p, C.c_void_p]
    other = Engine(0)
    try:
        freed = C.c_size_t(0)
        assert L.zpq_pool_trim(eng.ctx, C.byref(freed)) == 0           # whatever earlier tests left idle
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert L.zpq_dev_alloc_pooled(eng.ctx, 3 << 20, C.byref(a)) == 0 and L.zpq_dev_alloc_pooled(eng.ctx, 5 << 20, C.byref(b)) == 0
        assert L.zpq_dev_alloc_pooled(other.ctx, 7 << 20, C.byref(c)) == 0
        assert L.zpq_dev_free_pooled(eng.ctx, a) == 0
        assert L.zpq_pool_trim(eng.ctx, C.byref(freed)) == 0 and (3 << 20) <= freed.value < (5 << 20)      # a went back, b is in use
        assert L.zpq_pool_trim(eng.ctx, C.byref(freed)) == 0 and freed.value == 0
        assert L.zpq_pool_trim(other.ctx, C.byref(freed)) == 0 and freed.value == 0                       # c is in use
        assert L.zpq_dev_free_pooled(other.ctx, c) == 0
        assert L.zpq_pool_trim(other.ctx, C.byref(freed)) == 0 and freed.value >= (7 << 20)
        # b is still usable, and a fresh request is served after the trim
        assert L.zpq_dev_memset(eng.ctx, b, 1, 5 << 20) == 0 and L.zpq_sync(eng.ctx) == 0
        assert L.zpq_dev_alloc_pooled(eng.ctx, 3 << 20, C.byref(a)) == 0
        assert L.zpq_dev_free_pooled(eng.ctx, a) == 0 and L.zpq_dev_free_pooled(eng.ctx, b) == 0
    finally:
        other.close()
