"""GPU parity for the suffix-array path (SURVEY.md rows a9 + a8/SA: what -m2 runs): zpq_suffix_array_dev against the
oracle / the real divsufsort, and LZ77 jobs with args[5]-args[0] >= 21 against the oracle's restatement of
LZBuffer::fill (bit-exact code streams), framed blocks through the decoder, and size-independent properties at
block sizes the CPU oracle does not finish in seconds."""
import numpy as np
import pytest

import datagen
import orc
from test_sa_cpu import CASES

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_suffix_array_equals_oracle(eng, name, data):
    sa, isa = eng.suffix_array(data, inverse=True)
    assert np.array_equal(sa, orc.suffix_array(data))
    if len(data):
        assert np.array_equal(isa[sa], np.arange(len(data), dtype=np.uint32))


@needs_ref
def test_suffix_array_equals_divsufsort_4mib(eng):
    t = datagen.text_like(1 << 20, 31)
    data = t + datagen.mixed(2 << 20, 32) + t[: 1 << 19] + bytes(1 << 19)       # a long repeat and a run of zeros
    assert np.array_equal(eng.suffix_array(data), orc.ref_divsufsort(data))


def test_suffix_array_without_inverse_and_twice(eng):
    data = datagen.binary_like(300001, 7)
    a = eng.suffix_array(data)
    assert np.array_equal(a, eng.suffix_array(data)) and np.array_equal(a, orc.suffix_array(data))


SA_ARGS = [
    (0, 1, 4, 0, 7, 21, 1),      # method 2 on a block of up to 1 MiB
    (2, 1, 4, 0, 7, 23, 1),
    (6, 1, 4, 0, 7, 27, 1),      # method 2 at its default 64 MiB block size (rb = 2)
    (0, 1, 5, 0, 3, 21, 0),      # no lookahead, 7 neighbours
    (0, 1, 4, 0, 0, 21, 1),      # no neighbours at all: literals only
    (1, 1, 6, 0, 9, 23, 1),      # 511 neighbours
]


@pytest.mark.parametrize("args", SA_ARGS)
def test_sa_parse_equals_oracle(eng, args):
    """(the 511-neighbour case leaves the workgroup's LDS tile: the scan then reads the global arrays)"""
    blocks = [d for _, d in CASES]
    out = eng.lz77_encode(blocks, [args] * len(blocks))
    for (name, d), o in zip(CASES, out):
        assert o == orc.lz77_sa_encode(d, args), name


@pytest.mark.parametrize("seg", ["16", "64", "1000", "4096"])
def test_sa_parse_does_not_depend_on_the_speculation_segments(eng, seg, monkeypatch):
    """The chain is walked speculatively per segment and stitched: tiny segments force joins, own steps of the stitcher,
    matches that jump over whole segments and literal runs that cross many of them."""
    monkeypatch.setenv("ZPQ_SA_SEG", seg)
    args = (0, 1, 4, 0, 7, 21, 1)
    blocks = [d for _, d in CASES]
    out = eng.lz77_encode(blocks, [args] * len(blocks))
    for (name, d), o in zip(CASES, out):
        assert o == orc.lz77_sa_encode(d, args), name


def test_sa_parse_window_boundary(eng):
    t = datagen.text_like(100000, 21)
    data = t + t + t[:100000]                       # crosses two 2^17 windows at args[0] = 0
    args = (0, 1, 4, 0, 7, 21, 1)
    assert eng.lz77_encode([data], [args])[0] == orc.lz77_sa_encode(data, args)


def test_sa_and_hash_jobs_in_one_call(eng):
    a, b, c = datagen.mixed(400000, 1), datagen.text_like(300000, 2), datagen.binary_like(200000, 3)
    sa_args, ht_args = (0, 1, 4, 0, 7, 21, 1), (4, 1, 4, 0, 3, 24)
    out = eng.lz77_encode([a, b, c], [sa_args, ht_args, sa_args])
    assert out[0] == orc.lz77_sa_encode(a, sa_args)
    assert out[1] == orc.lz77_encode(b, ht_args)
    assert out[2] == orc.lz77_sa_encode(c, sa_args)


@needs_ref
def test_sa_parse_equals_reference_lzbuffer_8mib(eng):
    """An 8 MiB block (the real reference takes a few seconds): repeats longer than maxMatch, 4096+ literal runs."""
    t = datagen.text_like(3 << 20, 41)
    data = t + datagen.random_bytes(20000, 42) + t[1 << 20:] + datagen.mixed(3 << 20, 43) + bytes(70000)
    data = data[: 8 << 20]
    args = (3, 1, 4, 0, 7, 24, 1)
    assert eng.lz77_encode([data], [args])[0] == orc.ref_lzbuffer(data, args)


def test_method2_blocks_roundtrip_and_reference_decode(eng):
    blocks = [datagen.text_like(900000, 5), datagen.mixed(1 << 20, 6), b"", b"abc", bytes(50000)]
    methods = ["x0,1,4,0,7,21,1", "x0,1,4,0,7,21,1", "x0,1,4,0,7,21,1", "x0,1,4,0,7,21,1", "x6,1,4,0,7,27,1"]
    framed = eng.compress_blocks(blocks, methods, None, None, True)
    assert all(st == 0 for st, _ in framed)
    res = eng.decompress_blocks_resident([f for _, f in framed], [len(b) + 64 for b in blocks])
    for b, r in zip(blocks, res):
        assert r["status"] == 0 and r["data"] == b
    if orc.have_ref():
        for b, (_, f) in zip(blocks, framed):
            assert orc.ref_decompress(f, len(b) + 64) == b


def test_block_64mib_properties(eng):
    """At method 2's real block size the oracle is too slow: the stream must decode to the input (GPU decoder and the
    oracle's decoder), and the suffix array must be a permutation in suffix order (checked on a sample of neighbours)."""
    n = 64 << 20
    unit = datagen.text_like(4 << 20, 51) + datagen.mixed(4 << 20, 52)
    data = (unit * 8)[:n]                              # 8 MiB period: matches at distance 2^23, maxMatch-long
    args = (6, 1, 4, 0, 7, 27, 1)
    lz = eng.lz77_encode([data], [args])[0]
    assert len(lz) < n // 4
    assert orc.lz77_decode(lz, n, rb=2) == data
    small = data[: 3 << 20]
    sa = eng.suffix_array(small)
    assert np.array_equal(np.sort(sa), np.arange(len(small), dtype=np.uint32))
    rng = np.random.default_rng(5)
    for j in rng.integers(1, len(small), 2000):
        a, b = int(sa[j - 1]), int(sa[j])
        assert small[a:a + 70000] < small[b:b + 70000] or small[a:] < small[b:]


def test_jidac_add_with_method_2(eng):
    """add -m2 through the journaling shim: 64 MiB d blocks (one here), suffix-array LZ77 inside; the archive extracts
    to the files and the real reference decoder walks it."""
    from zpaqfranz_amd import engine as E
    shared = datagen.text_like(3 << 20, 81)
    files = [("t/one", shared + datagen.mixed(2 << 20, 82)), ("t/two", datagen.binary_like(1 << 20, 83)), ("t/three", shared), ("t/empty", b"")]
    arc, st = E.jidac_add(eng, b"", files, 20240101120000, method="2")
    total = sum(len(d) for _, d in files)
    assert st["d_blocks"] == 1 and (6 << 20) <= st["unique_bytes"] < total - (2 << 20)           # one 64 MiB block; the shared text (all but its edge fragments) once
    assert E.jidac_extract(eng, arc) == dict(files)
    arc14, _ = E.jidac_add(eng, b"", files, 20240101120000, method="14")
    assert len(arc) < len(arc14)                               # the longer search pays on this data
    if orc.have_ref():
        out = orc.ref_decompress(arc, 64 << 20)                # c, d, h, i blocks concatenated by libzpaq::decompress
        assert files[0][1] in out and files[1][1] in out          # the new fragments of a file sit in the d block in order


def test_text_m2_two_ranks(tmp_path):
    """bench.py --workload text_m2 under torch.distributed.run with two ranks (gloo through the host, both on GPU 0):
    blocks are independent, every rank compresses its own text, rank 0 reports the sum."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "text_m2", "--text-bytes", "5000000",
           "--steps", "1", "--warmup", "0", "--dist-backend", "gloo", "--same-device", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["input_bytes"] == 10000000 and d["scaling"] == "weak"
    assert 0.2 < d["config"]["ratio"] < 0.7 and d["value"] > 0


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_bwt_equals_oracle(eng, name, data):
    assert eng.bwt(data) == orc.bwt_encode(data)


@needs_ref
def test_bwt_equals_lzbuffer_level3_2mib(eng):
    data = datagen.mixed(2 << 20, 77)
    assert eng.bwt(data) == orc.ref_lzbuffer(data, (1, 3, 0, 0, 0, 0, 0))


L2_ARGS = [
    (0, 2, 12, 0, 7, 21, 1),     # what method 3 puts in front of its model (type < 640, not text)
    (1, 2, 5, 0, 7, 22, 1),      # method 4, types 24..47
    (0, 2, 1, 0, 4, 21, 0),      # shortest matches the byte codes allow
    (6, 2, 12, 0, 7, 27, 1),     # at the 64 MiB block size
]


@pytest.mark.parametrize("args", L2_ARGS)
def test_sa_parse_level2_byte_codes_equal_oracle(eng, args):
    """Byte-aligned codes (level 2): literal runs of at most 64 behind a length byte, matches in pieces of
    minMatch .. minMatch+63 with 2/3/4 offset bytes; a far match must be longer to be taken."""
    blocks = [d for _, d in CASES]
    out = eng.lz77_encode(blocks, [args] * len(blocks))
    for (name, d), o in zip(CASES, out):
        assert o == orc.lz77_sa_encode(d, args), name


def test_levels_mixed_in_one_call(eng):
    a, b = datagen.mixed(300000, 11), datagen.text_like(200000, 12)
    out = eng.lz77_encode([a, b, a], [(0, 2, 12, 0, 7, 21, 1), (0, 1, 4, 0, 7, 21, 1), (4, 1, 5, 0, 3, 24)])
    assert out[0] == orc.lz77_sa_encode(a, (0, 2, 12, 0, 7, 21, 1))
    assert out[1] == orc.lz77_sa_encode(b, (0, 1, 4, 0, 7, 21, 1))
    assert out[2] == orc.lz77_encode(a, (4, 1, 5, 0, 3, 24))
