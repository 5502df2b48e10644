"""Blocks of several segments (a streaming archive with more than one file per block): the decoder's model and the post-processor
carry on from segment to segment -- Decompresser::decompress initialises them for the first one only (ZSFX/libzpaq.cpp:2307-2337),
the Encoder's Predictor is initialised by startBlock.  The measure is the real reference: blocks coded with its Predictor kept
across the segments (oracle/_ref: ref_cm_encode_segments) must decode on the device, and blocks the shim's Compressor writes
must decode with the real libzpaq::decompress()."""
import ctypes as C
import hashlib
import os
import subprocess

import pytest

import cmconfigs
import datagen
import orc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])

# a post-processor with memory: OUT = input + the previous OUT; `c` is NOT cleared at the end of a segment, so what a
# later segment decodes to depends on where the one before it ended
DELTA_CFG = """comp 0 0 0 0 1
  0 icm 5
hcomp
  halt
pcomp delta ;
  a> 255 if halt endif
  a+=c c=a out
  halt
end
"""


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from zpaqfranz_amd import build
    here = build.HERE                      # (the product's directory; the emulated engine's when the suite runs on the CPU)
    drv = str(tmp_path_factory.mktemp("segdrv") / "segments_driver")
    if os.environ.get("ZPQ_EMU_ASAN") == "1":
        pytest.skip("a g++-linked driver does not link against the AddressSanitizer build of the emulated engine")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "zpaqfranz_amd", "shim"), os.path.join(ROOT, "tests", "cpp", "segments_driver.cpp"),
                           "-L" + here, "-lzpaq_gpu", "-lzpaqhip", "-Wl,-rpath," + here, "-o", drv])
    return drv


def _files():
    return [("a.txt", datagen.text_like(5000, 61)), ("b.bin", datagen.binary_like(3000, 62)), ("empty", b""), ("c.txt", datagen.text_like(2500, 61)),
            ("d.mix", datagen.mixed(4000, 63))]


def _frame(header, segs, coded, shas=True, stored=False):
    """a block as Compressor writes it: tag, zPQ level 1|2 type 1, header, then per segment 1 name 0 comment 0 0 data 253 sha1 | 254; 255"""
    out = bytearray(TAG + b"zPQ" + bytes([1 if header[6] else 2, 1]) + header)
    for (name, data), c in zip(segs, coded):
        out += b"\x01" + name.encode() + b"\0" + str(len(data)).encode() + b"\0\0"
        if stored:
            for p in range(0, len(c), 65536):
                out += len(c[p:p + 65536]).to_bytes(4, "big") + c[p:p + 65536]
            out += bytes(4)
        else:
            out += c
        out += (b"\xfd" + hashlib.sha1(data).digest()) if shas else b"\xfe"
    return bytes(out + b"\xff")


def _unblock(eng, blk, cap, nseg, verify=1):
    from zpaqfranz_amd.engine import UnblockJob
    L = eng.L
    job = (UnblockJob * 1)()
    src = C.create_string_buffer(blk + bytes(64), len(blk) + 64)
    out = C.create_string_buffer(cap + 64)
    ends = (C.c_uint32 * max(1, nseg))()
    job[0].in_ = C.cast(src, C.c_void_p); job[0].n = len(blk)
    job[0].out = C.cast(out, C.c_void_p); job[0].out_cap = cap
    job[0].seg_cap = nseg; job[0].seg_out_end = ends
    L.zpq_decompress_blocks.argtypes = [C.c_void_p, C.POINTER(UnblockJob), C.c_size_t, C.c_int]
    rc = L.zpq_decompress_blocks(eng.ctx, job, 1, verify)
    cuts = [0] + [int(e) for e in ends[:nseg]]
    return rc, job[0], [out.raw[cuts[i]:cuts[i + 1]] for i in range(nseg)] if job[0].status == 0 else None


@pytest.mark.parametrize("model", ["mid", "alltypes"])
def test_block_coded_by_the_reference_predictor_across_segments_decodes_on_the_device(eng, model):
    """The block is put together here from the REAL Predictor's output (kept across the segments): every segment's bytes come
    back, every stored SHA-1 is compared with its own segment, both through the C ABI."""
    h = orc.ref_compile(cmconfigs.ALL[model], [0] * 9)[0]
    files = _files()
    streams = [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)]        # the first one opens with the PASS byte
    coded = orc.ref_cm_encode_segments(h, streams)
    blk = _frame(h, files, coded)
    total = sum(len(d) for _, d in files)
    assert orc.ref_decompress(blk, total + 64) == b"".join(d for _, d in files)        # the reference agrees that this is a block
    rc, job, parts = _unblock(eng, blk, total + 64, len(files))
    assert rc == 0 and job.status == 0 and job.nseg == len(files) and job.consumed == len(blk)
    assert parts == [d for _, d in files]
    assert bytes(job.sha1) == hashlib.sha1(files[0][1]).digest()
    # a wrong stored checksum in the THIRD of five segments is found
    bad = bytearray(blk)
    at = blk.index(hashlib.sha1(files[3][1]).digest())
    bad[at + 5] ^= 1
    rc, job, _ = _unblock(eng, bytes(bad), total + 64, len(files))
    assert job.status == -7 and rc == -7                                                # ZPQ_ERR_CHECKSUM
    # without checksums, and cut short inside the second segment
    rc, job, parts = _unblock(eng, _frame(h, files, coded, shas=False), total + 64, len(files))
    assert rc == 0 and parts == [d for _, d in files]
    rc, job, _ = _unblock(eng, blk[:len(blk) // 2], total + 64, len(files))
    assert rc != 0 and job.status != 0


def test_stored_block_with_a_post_processor_that_remembers(eng):
    """No model, but a PCOMP program whose state crosses the segment borders: the ZPAQL machine runs on over all the
    segments, with the end-of-segment input after each (PostProcessor::write, ZSFX/libzpaq.cpp:2185-2226)."""
    hdr, pc = orc.ref_compile(DELTA_CFG.replace("comp 0 0 0 0 1\n  0 icm 5", "comp 0 0 0 0 0"), [0] * 9)      # (pc: two size bytes, then the program)
    assert hdr[6] == 0
    files = _files()
    prev = 0
    streams = []
    for i, (_, d) in enumerate(files):
        t = bytearray()
        for x in d:
            t.append((x - prev) & 255); prev = x
        streams.append((b"\x01" + pc if i == 0 else b"") + bytes(t))
    blk = _frame(hdr, files, streams, stored=True)
    total = sum(len(d) for _, d in files)
    assert orc.ref_decompress(blk, total + 64) == b"".join(d for _, d in files)
    rc, job, parts = _unblock(eng, blk, total + 64, len(files))
    assert rc == 0 and job.status == 0 and parts == [d for _, d in files]
    # the same framing without a program (0 in front of the first segment): every segment is a copy of its stored bytes
    blk = _frame(hdr, files, [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)], stored=True)
    assert orc.ref_decompress(blk, total + 64) == b"".join(d for _, d in files)
    rc, job, parts = _unblock(eng, blk, total + 64, len(files))
    assert rc == 0 and job.status == 0 and job.nseg == len(files) and parts == [d for _, d in files]
    # ... and too little room for the third segment
    rc, job, _ = _unblock(eng, blk, len(files[0][1]) + len(files[1][1]) + 10, len(files))
    assert job.status == -4 or rc == -4                                                  # ZPQ_ERR_CAPACITY


def test_shim_compressor_and_decompresser_classes_over_blocks_of_segments(eng, driver, tmp_path):
    """A streaming archiver's loop -- startBlock, then startSegment / compress / endSegment per file, endBlock -- through the
    shim: what it writes is read back by the real libzpaq::decompress(), and by the shim's Decompresser segment by segment
    (names, sizes, stored and computed SHA-1 per segment).  Built-in model 2 (no post-processor) and a model + a post-processor
    with memory."""
    files = _files()
    paths = []
    for name, d in files:
        p = tmp_path / name
        p.write_bytes(d)
        paths.append(str(p))
    cfg = tmp_path / "delta.cfg"
    cfg.write_text(DELTA_CFG)
    whole = b"".join(d for _, d in files)
    for how, pre in (("2", "none"), ("@" + str(cfg), "delta")):
        arc = tmp_path / ("s%s.zpaq" % pre)
        r = subprocess.run([driver, "c", str(arc), how, pre] + paths, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        blk = arc.read_bytes()
        assert orc.ref_decompress(blk, len(whole) + 64) == whole, how               # the real Decompresser, model kept across segments
        out = tmp_path / ("out" + pre)
        out.mkdir()
        r = subprocess.run([driver, "d", str(arc), str(out)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        lines = [l.split(" ") for l in r.stdout.strip().splitlines()]
        assert [l[2] for l in lines] == [n for n, _ in files]
        for k, (name, d) in enumerate(files):
            assert lines[k][:2] == ["0", str(k)] and int(lines[k][3]) == len(d)
            assert lines[k][4] == lines[k][5] == hashlib.sha1(d).hexdigest(), name
            assert (out / ("0.%d" % k)).read_bytes() == d
    # the second segment is not what a fresh model would have made of it
    h = orc.ref_compile(cmconfigs.ALL["mid"], [0] * 9)[0]
    two = orc.ref_cm_encode_segments(h, [b"\0" + files[0][1], files[1][1]])
    assert two[1] != orc.ref_cm_encode(h, files[1][1])


def test_damaged_blocks_of_segments_end_in_a_status(eng):
    """Flipped bytes, cuts, a stored length that runs past the block (and past 2^32 when added to its position), a segment count
    beyond the limit: every one ends in a status -- or, where the damage hit a name or a comment, in the right bytes -- never in
    reads or writes outside the buffers (the emulated engine runs this under AddressSanitizer: tools/emu/asan.sh)."""
    import numpy as np
    rng = np.random.default_rng(77)
    files = _files()
    total = sum(len(d) for _, d in files)
    want = [d for _, d in files]
    h = orc.ref_compile(cmconfigs.ALL["mid"], [0] * 9)[0]
    coded = orc.ref_cm_encode_segments(h, [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)])
    hdr, pc = orc.ref_compile(DELTA_CFG.replace("comp 0 0 0 0 1\n  0 icm 5", "comp 0 0 0 0 0"), [0] * 9)
    prev = 0
    streams = []
    for i, (_, d) in enumerate(files):
        t = bytearray()
        for x in d:
            t.append((x - prev) & 255); prev = x
        streams.append((b"\x01" + pc if i == 0 else b"") + bytes(t))
    for blk in (_frame(h, files, coded), _frame(hdr, files, streams, stored=True)):
        for trial in range(40):
            b = bytearray(blk)
            kind = trial % 4
            if kind == 0:
                for _ in range(1 + trial % 3):
                    b[int(rng.integers(13, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b = b[:int(rng.integers(20, len(b)))]
            elif kind == 2:
                at = int(rng.integers(13, len(b) - 4))
                b[at:at + 4] = bytes([255, 255, 255, int(rng.integers(0, 256))])
            else:
                at = int(rng.integers(13, len(b)))
                b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8))
            rc, job, parts = _unblock(eng, bytes(b), total + 64, len(files))
            assert rc != 0 or parts == want or job.nseg != len(files), (trial, kind)
    # more segments than the engine takes in one block
    many = [("f", b"")] * 70000
    blk = _frame(hdr, many, [b"\0"] + [b""] * 69999, shas=False, stored=True)
    rc, job, _ = _unblock(eng, blk, 64, 1)
    assert rc != 0 and job.status != 0


def test_shim_decompresser_on_damaged_archives_of_segments(eng, driver, tmp_path):
    """The shim reads a block of continuing segments to its end before it decodes the first one: cut, flipped and padded archives
    end in libzpaq::error() (exit code 1 of the driver) or in segments whose checksums the caller can compare -- not in a crash
    and not in a wait for bytes that never come."""
    import numpy as np
    rng = np.random.default_rng(5)
    files = _files()
    h = orc.ref_compile(cmconfigs.ALL["mid"], [0] * 9)[0]
    coded = orc.ref_cm_encode_segments(h, [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)])
    blk = _frame(h, files, coded)
    out = tmp_path / "o"
    out.mkdir()
    for trial in range(12):
        b = bytearray(blk)
        if trial % 3 == 0:
            b = b[:int(rng.integers(30, len(b) - 1))]
        elif trial % 3 == 1:
            b[int(rng.integers(13, len(b)))] ^= 1 << int(rng.integers(0, 8))
        else:
            at = int(rng.integers(13, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, size=5, dtype=np.uint8))
        arc = tmp_path / ("bad%d.zpaq" % trial)
        arc.write_bytes(bytes(b))
        r = subprocess.run([driver, "d", str(arc), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode in (0, 1), (trial, r.returncode, r.stderr[-300:])
        if r.returncode == 0:                     # decoded: whatever differs shows in the checksums the driver prints
            for l in r.stdout.strip().splitlines():
                f = l.split(" ")
                assert len(f) == 6


def test_shim_findblock_searches_like_the_reference(eng, driver, tmp_path):
    """Decompresser::findBlock as the reference searches (ZSFX/libzpaq.cpp:2239-2262): four rolling hashes over the last 16
    bytes, pre-seeded with the tag -- (i) a stream written without Compressor::writeTag() (it begins with "zPQ") is a block,
    (ii) a tag that is not followed by "zPQ" is just bytes and the scan goes on to the real block behind it, (iii) several
    blocks, the second without its tag but behind bytes that end in the tag.  The real libzpaq::decompress() reads the same
    streams to the same bytes."""
    files = _files()
    h = orc.ref_compile(cmconfigs.ALL["mid"], [0] * 9)[0]
    coded = orc.ref_cm_encode_segments(h, [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)])
    blk = _frame(h, files, coded)
    assert blk[:13] == TAG and blk[13:16] == b"zPQ"
    one = _frame(h, files[:1], [orc.ref_cm_encode(h, b"\0" + files[0][1])])
    lookalike = b"garbage" + TAG + b"zPx" + bytes(range(40)) + TAG[:12] + b"\x00zPQ" + TAG + TAG[:5]
    cases = {
        "notag": (blk[13:], [files]),
        "lookalike": (lookalike + blk, [files]),
        # the second block sits behind bytes that end in the tag; the third has no tag and no tag in front of it: never found
        "two": (one + b"junk" + TAG[3:] + TAG + one[13:] + lookalike + blk[13:], [files[:1], files[:1]]),
        "two_b": (blk[13:] + lookalike + TAG + one[13:], [files, files[:1]]),
    }
    for name, (arc_bytes, want_blocks) in cases.items():
        whole = b"".join(d for fs in want_blocks for _, d in fs)
        assert orc.ref_decompress(arc_bytes, len(whole) + 64) == whole, name              # what the real reference makes of the stream
        arc = tmp_path / (name + ".zpaq")
        arc.write_bytes(arc_bytes)
        out = tmp_path / ("o_" + name)
        out.mkdir()
        r = subprocess.run([driver, "d", str(arc), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr)
        lines = [l.split(" ") for l in r.stdout.strip().splitlines()]
        want = [(b, s, n, d) for b, fs in enumerate(want_blocks) for s, (n, d) in enumerate(fs)]
        assert len(lines) == len(want), (name, r.stdout)
        for l, (b, s, n, d) in zip(lines, want):
            assert l[:3] == [str(b), str(s), n] and int(l[3]) == len(d) and l[4] == l[5] == hashlib.sha1(d).hexdigest(), (name, l)
    # a level byte that is neither 1 nor 2 behind "zPQ" is an error, not a reason to keep scanning (ZSFX/libzpaq.cpp:2256)
    arc = tmp_path / "level.zpaq"
    arc.write_bytes(b"zPQ\x03\x01" + blk[18:])
    r = subprocess.run([driver, "d", str(arc), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "level" in r.stderr


def test_shim_delivers_the_whole_segments_in_front_of_a_damaged_tail(eng, driver, tmp_path):
    """A block of continuing segments cut inside its third segment (or with a wrong byte where the fourth should begin): the
    reference decodes segment by segment and fails where the damage is (findFilename / the decoder), so the segments in front
    arrive.  The shim reads ahead to decode the block in one device job -- it must still deliver them, and report the damage
    when the caller gets there."""
    files = _files()
    h = orc.ref_compile(cmconfigs.ALL["mid"], [0] * 9)[0]
    coded = orc.ref_cm_encode_segments(h, [(b"\0" if i == 0 else b"") + d for i, (_, d) in enumerate(files)])
    blk = _frame(h, files, coded)
    # where every segment starts: 1 name 0 comment 0 0
    starts, at = [], 13 + 5 + len(h)
    for (name, data), c in zip(files, coded):
        starts.append(at)
        at += 1 + len(name) + 1 + len(str(len(data))) + 2 + len(c) + 21
    assert blk[at] == 255 and all(blk[s] == 1 for s in starts)
    cut3 = blk[:starts[3] + 9]                               # ends inside the fourth segment's header
    cut2 = blk[:starts[2] + 12 + len(coded[2]) // 2]         # ends inside the third segment's coded bytes
    wrong = bytearray(blk); wrong[starts[3]] = 7             # neither a segment nor the end of the block
    for name, arc_bytes, whole in (("cut3", cut3, 3), ("cut2", cut2, 2), ("wrong", bytes(wrong), 3)):
        arc = tmp_path / (name + ".zpaq")
        arc.write_bytes(arc_bytes)
        out = tmp_path / ("o_" + name)
        out.mkdir()
        r = subprocess.run([driver, "d", str(arc), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 1, (name, r.returncode, r.stdout, r.stderr)
        lines = [l.split(" ") for l in r.stdout.strip().splitlines()]
        assert len(lines) == whole, (name, r.stdout, r.stderr)
        for k, l in enumerate(lines):
            n, d = files[k]
            assert l[:3] == ["0", str(k), n] and l[4] == l[5] == hashlib.sha1(d).hexdigest(), (name, l)
