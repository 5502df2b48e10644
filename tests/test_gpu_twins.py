"""Twin files (zpaqfranz_amd/csrc/twins.hip): whole-file duplicates are found by comparing bytes on the device, only the
representatives are fragmented and hashed, the twins take their records.  The fused call must give exactly what the
fragment loop + SHA-1 give for every file on its own (the oracle), whatever is duplicated and however it is aligned."""
import struct

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


def run(eng, files, twins=True, params=None):
    """-> (records [(file, offset in file, length, sha1)], rep, stats) through zpq_fragment_sha1_dev"""
    params = params or eng.fragment_params()
    file_off = [0]
    for f in files:
        file_off.append(file_off[-1] + len(f))
    data = eng.upload(b"".join(files) + bytes(64))
    cap = max(1, eng.fragment_capacity(file_off, params))
    fo, fl, ff, dg = eng.alloc(cap * 8), eng.alloc(cap * 4), eng.alloc(cap * 4), eng.alloc(cap * 20 + 64)
    try:
        n, rep, st = eng.fragment_sha1_dev(data.ptr, file_off, params, fo.ptr, fl.ptr, ff.ptr, dg.ptr, cap, twins=twins, want_rep=True)
        eng.sync()
        offs = struct.unpack("<%dQ" % n, fo.download(n * 8))
        lens = struct.unpack("<%dI" % n, fl.download(n * 4))
        fil = struct.unpack("<%dI" % n, ff.download(n * 4))
        dig = dg.download(n * 20)
    finally:
        for b in (data, fo, fl, ff, dg):
            b.free()
    return [(fil[i], offs[i] - file_off[fil[i]], lens[i], dig[20 * i:20 * i + 20]) for i in range(n)], rep, st


def expect(files, params=None):
    kw = {} if params is None else dict(fragment=params.fragment_log2, min_frag=params.min_fragment, max_frag=params.max_fragment)
    out = []
    for fi, f in enumerate(files):
        off = 0
        for ln in orc.chunk(f, **kw):
            out.append((fi, off, ln, orc.sha1(f[off:off + ln])))
            off += ln
    return out


def expected_rep(files, min_bytes=4096):
    rep = []
    for i, f in enumerate(files):
        r = i
        if len(f) >= min_bytes:
            for j in range(i):
                if len(files[j]) == len(f) and files[j] == f:
                    r = j
                    break
        rep.append(r)
    return rep


def flip(b, pos):
    a = bytearray(b)
    a[pos] ^= 0x40
    return bytes(a)


def test_twins_of_every_alignment_and_near_twins(eng):
    """Copies at every offset mod 16 (odd-length spacers in between), files that differ from their would-be representative in
    the first byte (the unaligned head), the last byte (the tail), around a chunk edge and in the middle, equal lengths
    with other contents, short files and empty files."""
    a = datagen.mixed(700001, 1)            # > 2 chunks of 256 KiB, odd length
    b = datagen.text_like(300000, 2)
    c = datagen.binary_like(5000, 3)
    files = [a, b, b"", c]
    for k in range(17):
        files.append(datagen.random_bytes(k + 1, 40 + k))       # spacer: shifts the alignment of what follows
        files.append(a if k % 2 else b)
    files += [flip(a, 0), flip(a, len(a) - 1), flip(a, 262144), flip(a, 262143 + 16), flip(a, 350000), flip(a, 7), flip(a, 15), flip(a, 16)]
    files += [datagen.mixed(700001, 9), a, c, c, b"", datagen.random_bytes(100, 5), datagen.random_bytes(100, 5), b[:-1], b]
    got, rep, st = run(eng, files)
    assert rep == expected_rep(files)
    assert st["twins"] == sum(1 for i, r in enumerate(rep) if r != i)
    assert got == expect(files)
    plain, rep0, st0 = run(eng, files, twins=False)
    assert plain == got and rep0 == list(range(len(files))) and st0["twins"] == 0


def test_many_copies_of_a_corpus(eng):
    """The shape of the headline workload: a small corpus replicated 9 times back to back."""
    corpus = [b for _, b in datagen.silesia_like(seed=3, scale=0.004)]
    files = corpus * 9
    got, rep, st = run(eng, files)
    assert rep == [i % len(corpus) for i in range(len(files))]
    assert st["twins"] == 8 * len(corpus) and st["twin_bytes"] == 8 * sum(len(b) for b in corpus)
    assert got == expect(files)


def test_nothing_to_fold(eng):
    files = [datagen.mixed(200000 + 1000 * i, 20 + i) for i in range(5)] + [datagen.mixed(200000, 30)]   # one equal length, other bytes
    got, rep, st = run(eng, files)
    assert rep == list(range(len(files))) and st["twins"] == 0
    assert got == expect(files)


def test_same_fingerprint_samples_but_different_bytes(eng):
    """Files that agree on every sampled position and differ elsewhere are compared in full and kept apart; a second
    class of equal files with the same length gets its own representative."""
    base = datagen.mixed(400000, 7)
    n = len(base)
    sample = {(n - 16) // 7 * i + d for i in range(7) for d in range(16)} | {n - 16 + d for d in range(16)}
    pos = next(p for p in range(1000, n) if p not in sample)
    other = flip(base, pos)
    files = [base, other, base, other, flip(base, pos + 1)]
    got, rep, st = run(eng, files)
    # 1 and 3 share the fingerprint of 0 and are compared with it: they differ, so they stay on their own
    assert rep == [0, 1, 0, 3, 4]
    assert got == expect(files)


def test_other_fragment_settings_and_the_twin_api(eng):
    p = eng.fragment_params(fragment=4, min_fragment=1024, max_fragment=130048)
    a = datagen.mixed(90000, 11)
    files = [a, datagen.random_bytes(3, 1), a, a[:50000], a]
    got, rep, _ = run(eng, files, params=p)
    assert rep == [0, 1, 0, 3, 0]
    assert got == expect(files, p)
    file_off = [0]
    for f in files:
        file_off.append(file_off[-1] + len(f))
    data = eng.upload(b"".join(files) + bytes(64))
    try:
        rep2, st = eng.file_twins_dev(data.ptr, file_off, min_bytes=1)
    finally:
        data.free()
    assert rep2 == [0, 1, 0, 3, 0] and st["twins"] == 2 and st["compared"] == 2
