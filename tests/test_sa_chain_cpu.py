"""The parallel formulation of the suffix-array parse (lz77_sa.hip), modelled on the CPU and checked against the serial
restatement of LZBuffer::fill:

 1. the decision at a position is a function of (position, lit == 0) alone (orc_lz77_sa_decisions evaluates it for every
    position without ever parsing);
 2. between two positions reached with lit == 0 ("nodes": block start, end of a match, flush of a 4096-byte literal run)
    the parse is a function of the first one: take the lit==0 decision there, else the first lit>0 decision that takes
    within 4095 further positions, else flush;
 3. chains started at every segment start merge with the true chain at their first common node: one chain per segment
    plus a stitcher that adopts a segment's tokens where it lands on a visited node gives the serial token list, for any
    segment size."""
import numpy as np
import pytest

import datagen
import orc

MAXLIT = 4096


def next_node(rec, takeb, n, p):
    """(next node, token or None) from node p -- lz77_sa.hip next_node()."""
    r = int(rec[2 * p])
    m = p
    if r == 0:
        lim = min(p + MAXLIT, n)
        nz = np.flatnonzero(takeb[p + 1:lim])
        if len(nz) == 0:
            return lim, None
        m = p + 1 + int(nz[0])
        r = int(rec[2 * m + 1])
    blen, blit, off = r & 0xffff, (r >> 16) & 0xffff, r >> 32
    return m + blen, (m + blit, blen - blit, off)


def serial_chain(rec, takeb, n):
    toks, p = [], 0
    while p < n:
        p, t = next_node(rec, takeb, n, p)
        if t:
            toks.append(t)
    return toks


def stitched_chain(rec, takeb, n, seg):
    nseg = max(1, -(-n // seg))
    visit = np.zeros(n + 1, dtype=bool)
    stok, sexit = [[] for _ in range(nseg)], [0] * nseg
    for k in range(nseg):                                   # one speculative chain per segment, from its first position
        p, e = k * seg, min((k + 1) * seg, n)
        while p < e:
            visit[p] = True
            q, t = next_node(rec, takeb, n, p)
            if t:
                stok[k].append((p, t))
            p = q
        sexit[k] = p
    toks, p, own_steps = [], 0, 0
    joined = [False] * nseg
    while p < n:                                            # the true chain
        k = p // seg
        if not joined[k] and visit[p]:
            joined[k] = True
            toks += [t for node, t in stok[k] if node >= p]
            p = sexit[k]
            continue
        p, t = next_node(rec, takeb, n, p)
        own_steps += 1
        if t:
            toks.append(t)
    return toks, own_steps


CASES = [("text", datagen.text_like(50000, 3)), ("mixed", datagen.mixed(60000, 4)), ("random", datagen.random_bytes(20000, 5)),
         ("zeros", bytes(30000)), ("abab", b"ab" * 9000 + b"c"),
         ("repeats", datagen.text_like(9000, 8) * 3 + datagen.random_bytes(9000, 9) + datagen.text_like(9000, 8))]


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("args", [(0, 1, 4, 0, 7, 21, 1), (0, 1, 5, 0, 3, 21, 0), (6, 1, 4, 0, 7, 27, 1)])
def test_chain_over_precomputed_decisions_is_the_serial_parse(name, data, args):
    sa = orc.suffix_array(data)
    rec = orc.lz77_sa_decisions(data, args, sa)
    takeb = rec[1::2] != 0
    n = len(data)
    _, want = orc.lz77_sa_encode(data, args, sa=sa, trace=True)
    assert serial_chain(rec, takeb, n) == want
    for seg in (64, 1000, 8192):
        got, own = stitched_chain(rec, takeb, n, seg)
        assert got == want, (seg,)
        assert own <= len(want) + n // MAXLIT + 2
