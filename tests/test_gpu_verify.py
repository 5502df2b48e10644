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
