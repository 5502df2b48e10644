"""The LZ77 level-1 decoder has two forms with identical results: one wave that parses and copies (lz77_decode_kernel)
and the token path (speculative parse of 1 KiB stream segments, stitched; one wave replays the token list).  Both are
run on real, truncated, bit-flipped and random streams and must agree with each other and, where the stream is valid,
with the oracle's decoder."""
import numpy as np
import pytest

import datagen
import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def both(eng, monkeypatch, streams, caps, rb=0):
    monkeypatch.setenv("ZPQ_LZDEC_SERIAL", "1")
    a = eng.lz77_decode(streams, caps, rb=rb)
    monkeypatch.setenv("ZPQ_LZDEC_SERIAL", "0")
    b = eng.lz77_decode(streams, caps, rb=rb)
    return a, b


def test_token_path_equals_wave_decoder_on_real_streams(eng, monkeypatch):
    blocks = [datagen.mixed(3 << 20, 31), datagen.text_like(900000, 32), datagen.random_bytes(300000, 33), bytes(1 << 20), b"ab" * 300000,
              b"x", datagen.binary_like(2 << 20, 34)]
    streams = [orc.lz77_encode(b, [4, 1, 5, 0, 3, 24]) for b in blocks]
    a, b = both(eng, monkeypatch, streams, [len(x) + 64 for x in blocks])
    assert a == b and [(s, o) for s, o in b] == [(0, x) for x in blocks]


def test_token_path_blocks_with_raw_offset_bits(eng, monkeypatch):
    data = datagen.mixed(5 << 20, 41)
    for a0 in (5, 6):
        s = orc.ref_lzbuffer(data, (a0, 1, 5, 0, 3, 19 + a0 + 1)) if orc.have_ref() else None
        if s is None:
            pytest.skip("needs oracle/_ref")
        a, b = both(eng, monkeypatch, [s], [len(data) + 64], rb=a0 - 4)
        assert a == b and b[0] == (0, data)


def test_truncated_streams(eng, monkeypatch):
    data = datagen.mixed(1 << 20, 51)
    s = orc.lz77_encode(data, [4, 1, 5, 0, 3, 24])
    cuts = [1, 2, 3, 7, 8, 9, 100, 1023, 1024, 1025, 5000, len(s) - 1, len(s) - 2, len(s) // 2]
    streams = [s[:-c] for c in cuts]
    a, b = both(eng, monkeypatch, streams, [len(data) + 64] * len(streams))
    assert a == b
    for (st, out), t in zip(b, streams):
        assert st == 0 and out == orc.lz77_decode(t, len(data) + 64)


def test_damaged_and_random_streams_agree(eng, monkeypatch):
    """Whatever a damaged stream decodes to (or fails with), both forms do the same: same status, same bytes produced."""
    rng = np.random.default_rng(7)
    data = datagen.text_like(600000, 61)
    s = bytearray(orc.lz77_encode(data, [4, 1, 5, 0, 3, 24]))
    streams = []
    for k in range(24):
        t = bytearray(s)
        for p in rng.integers(0, len(t), 1 + k % 5):
            t[p] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(t))
    streams += [rng.integers(0, 256, int(sz), dtype=np.uint8).tobytes() for sz in (1, 2, 3, 17, 1024, 1025, 70000, 300000)]
    streams += [b"\xff" * 5000, b"\x00" * 5000, b"\x55" * 5000, b"\xaa" * 70000]
    cap = 4 << 20
    a, b = both(eng, monkeypatch, streams, [cap] * len(streams))
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], i
        assert x[1] == y[1], i


def test_capacity_and_offset_errors_keep_their_order(eng, monkeypatch):
    data = datagen.mixed(400000, 71)
    s = orc.lz77_encode(data, [4, 1, 5, 0, 3, 24])
    a, b = both(eng, monkeypatch, [s, s, s], [1000, 123457, len(data)])
    assert a == b and [x[0] for x in b] == [-4, -4, 0] and b[2][1] == data
    assert b[0][1] == data[: len(b[0][1])] and b[1][1] == data[: len(b[1][1])]


def test_many_streams_in_one_call(eng, monkeypatch):
    blocks = [datagen.mixed(50000 + 977 * k, 100 + k) for k in range(150)]
    streams = [orc.lz77_encode(b, [4, 1, 5, 0, 3, 24]) for b in blocks]
    a, b = both(eng, monkeypatch, streams, [len(x) + 64 for x in blocks])
    assert a == b and [o for _, o in b] == blocks


def test_more_than_65536_stream_segments_in_one_call():
    """11 blocks of 64 MiB of text through method 2 and back: 325 MB of code stream (79 000 4-KiB segments) in ONE decode
    call.  The token path splits such a call into groups of launches (see zpq_lz77_decode_launch); both decoders must
    restore every block.  (Own process: the text is generated with torch, which wants to start the HIP runtime itself.)"""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(orc.ROOT, "tests", "lzdec_big_roundtrip.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "roundtrip ok serial=1" in r.stdout and "roundtrip ok serial=0" in r.stdout
