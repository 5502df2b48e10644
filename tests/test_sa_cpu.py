"""Suffix-array path of LZBuffer (args[5]-args[0] >= 21: method 2's "x<N>,1,4,0,7,<21+N>,1"): the oracle's restatement
(oracle/zpaq_oracle.cpp: orc_suffix_array, orc_lz77_sa_encode) pinned against the REAL reference compiled in place --
divsufsort (ZSFX/libzpaq.cpp:6047) and LZBuffer::fill (:6329-6453) through oracle/_ref."""
import numpy as np
import pytest

import datagen
import orc

pytestmark = pytest.mark.ref


def corpus():
    rng = np.random.default_rng(11)
    yield "empty", b""
    yield "one", b"a"
    yield "aaaa", b"a" * 5000
    yield "abab", b"ab" * 3000 + b"b"
    yield "zeros_tail", b"\0" * 9 + b"abc\0\0\0" + b"\0" * 7
    yield "text", datagen.text_like(60000, 3)
    yield "mixed", datagen.mixed(90000, 4)
    yield "binary", datagen.binary_like(50000, 5)
    yield "random", rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    t = datagen.text_like(20000, 8)
    yield "repeats", t + t[:15000] + t[3000:] + t          # long matches across the 4096-literal and 255 limits
    yield "two_symbols", rng.integers(0, 2, 40000, dtype=np.uint8).tobytes()


CASES = list(corpus())


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_suffix_array_equals_divsufsort(name, data):
    assert np.array_equal(orc.suffix_array(data), orc.ref_divsufsort(data))


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("args", [
    (0, 1, 4, 0, 7, 21, 1),      # method 2 on a block of up to 1 MiB
    (2, 1, 4, 0, 7, 23, 1),
    (6, 1, 4, 0, 7, 27, 1),      # method 2 at its default 64 MiB block size: rb = 2 raw offset bits
    (0, 1, 5, 0, 3, 21, 0),      # no lookahead, 7 neighbours
    (0, 1, 4, 0, 2, 22, 2),      # two bytes of lookahead
    (0, 2, 12, 0, 7, 21, 1),     # level 2 (byte codes) as methods 3 and 4 use it in front of a model
    (1, 2, 5, 0, 7, 22, 1),
    (0, 2, 1, 0, 4, 21, 0),
])
def test_sa_parse_equals_lzbuffer(name, data, args):
    assert orc.lz77_sa_encode(data, args) == orc.ref_lzbuffer(data, args)


def test_window_boundary_skips_the_lookahead():
    """isa[] holds one 2^(17+args[0]) window: at the last position of a window the lookahead candidate search is
    skipped (sa[isa[(i+1)&mask]] != i+1).  A 300 KiB input crosses two boundaries at args[0] = 0."""
    t = datagen.text_like(100000, 21)
    data = t + t + t[:100000]
    args = (0, 1, 4, 0, 7, 21, 1)
    assert orc.lz77_sa_encode(data, args) == orc.ref_lzbuffer(data, args)


def test_level1_stream_decodes_to_the_input():
    data = datagen.mixed(70000, 9)
    for a0 in (0, 6):
        lz = orc.lz77_sa_encode(data, (a0, 1, 4, 0, 7, 21 + a0, 1))
        assert orc.lz77_decode(lz, len(data), rb=max(a0 - 4, 0)) == data


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_bwt_output_equals_lzbuffer_level3(name, data):
    """LZBuffer with args[1] & 3 == 3 (what methods 3 and 4 put in front of their models) is the BWT of the block."""
    assert orc.bwt_encode(data) == orc.ref_lzbuffer(data, (0, 3, 0, 0, 0, 0, 0))


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_gpu_construction_model_equals_oracle(name, data):
    """tools/proto/sa_doubling.py: the steps of the GPU construction (8-byte keys, doubling rounds over the still
    ambiguous suffixes, length as second key for a suffix that ends first, singleton compaction) in numpy."""
    import os, sys
    sys.path.insert(0, os.path.join(orc.ROOT, "tools", "proto"))
    import sa_doubling
    sa, isa, rounds = sa_doubling.suffix_array(data)
    assert np.array_equal(sa, orc.suffix_array(data))
    if len(data):
        assert np.array_equal(isa[sa], np.arange(len(data), dtype=np.uint32))
