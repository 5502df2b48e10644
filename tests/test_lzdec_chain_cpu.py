"""The parallel formulation of the LZ77 level-1 decoder's PARSE (lz77_dec.hip), modelled on the CPU: the parse state
between two tokens is the bit position alone, so parses started anywhere (here: 256 bytes before every 4 KiB... here a
few hundred bits before every small segment) merge with the true parse at their first common token start; a stitcher
that adopts a segment's parse where it lands on a visited token start reproduces the serial token list, and replaying
that list reproduces the oracle decoder's output."""
import pytest

import datagen
import orc


def parse_token(s, nbits, rb, bp):
    """(kind, len, x, next) at bit bp: 'lit' (x = bit position of the bytes) / 'match' (x = offset) / 'end' / 'bad'."""
    def bits(b, k):
        v = 0
        for i in range(k):
            q = b + i
            v |= ((s[q >> 3] >> (q & 7)) & 1) << i if q < nbits else 0
        return v

    def gamma(b):
        v, used = 1, 0
        while bits(b + used, 1):
            v = v * 2 + bits(b + used + 1, 1)
            used += 2
            if used > 48:
                return None, None
        return v, used + 1
    if bp + 2 > nbits:
        return "end", 0, 0, None
    mm = bits(bp, 2)
    used = 2
    if mm == 0:
        ln, nb = gamma(bp + used)
        if ln is None:
            return ("bad" if bp + used + 50 <= nbits else "end"), 0, 0, None
        used += nb
        if bp + used > nbits:
            return "end", 0, 0, None
        b = bp + used
        avail = (nbits - b) >> 3
        cut = avail < ln
        ln = min(ln, avail)
        return "lit", ln, b, (None if cut else b + 8 * ln)
    if bp + 5 > nbits:
        return "end", 0, 0, None
    lo = (mm - 1) * 8 + bits(bp + 2, 3)
    used += 3
    g, nb = gamma(bp + used)
    if g is None:
        return ("bad" if bp + used + 50 <= nbits else "end"), 0, 0, None
    used += nb
    if bp + used + 2 > nbits:
        return "end", 0, 0, None
    ln = g * 4 + bits(bp + used, 2)
    used += 2
    b = bp + used
    if b + rb + lo > nbits:
        return "end", 0, 0, None
    r = bits(b, rb)
    q = bits(b + rb, lo) | (1 << lo)
    return "match", ln, ((q << rb) | r) - ((1 << rb) - 1), b + rb + lo


def serial_tokens(s, rb):
    nbits, p, out = len(s) * 8, 0, []
    while p is not None and p < nbits:
        k, ln, x, nxt = parse_token(s, nbits, rb, p)
        if k in ("end", "bad"):
            if k == "bad":
                out.append(("bad", 0, 0))
            break
        if ln:
            out.append((k, ln, x))
        p = nxt
    return out


def stitched_tokens(s, rb, seg_bits, warm_bits):
    nbits = len(s) * 8
    nseg = nbits // seg_bits + 1
    visit, sexit = set(), [None] * nseg
    for k in range(nseg):
        beg, end = k * seg_bits, min((k + 1) * seg_bits, nbits)
        p = max(0, beg - warm_bits) if k else 0
        while p is not None and p < end:
            if p >= beg:
                visit.add(p)
            kind, ln, x, nxt = parse_token(s, nbits, rb, p)
            if kind == "bad":
                p = "dead"; break
            p = nxt
        sexit[k] = p
    entry, p, own = [None] * nseg, 0, 0
    while p is not None and p != "dead" and p < nbits:          # the true chain: first token start of every segment
        k = p // seg_bits
        if entry[k] is None:
            entry[k] = p
        if p in visit:
            p = sexit[k]
            continue
        kind, ln, x, nxt = parse_token(s, nbits, rb, p)
        own += 1
        if kind in ("end", "bad"):
            break
        p = nxt
    out = []
    for k in range(nseg):                                        # every segment parsed again from its entry
        p, end = entry[k], min((k + 1) * seg_bits, nbits)
        while p is not None and p < end:
            kind, ln, x, nxt = parse_token(s, nbits, rb, p)
            if kind == "end":
                break
            if kind == "bad":
                out.append(("bad", 0, 0)); break
            if ln:
                out.append((kind, ln, x))
            p = nxt
    return out, own


def replay(s, toks):
    out = bytearray()
    for kind, ln, x in toks:
        if kind == "bad":
            return None
        if kind == "lit":
            b0, sh = x >> 3, x & 7
            for j in range(ln):
                two = s[b0 + j] | ((s[b0 + j + 1] << 8) if sh else 0)
                out.append((two >> sh) & 255)
        else:
            if x == 0 or x > len(out):
                return None
            for _ in range(ln):
                out.append(out[-x])
    return bytes(out)


@pytest.mark.parametrize("seed,kind", [(1, "mixed"), (2, "text"), (3, "binary"), (4, "random")])
def test_stitched_parse_equals_serial_parse_and_decodes(seed, kind):
    data = {"mixed": datagen.mixed, "text": datagen.text_like, "binary": datagen.binary_like, "random": datagen.random_bytes}[kind](24000, seed)
    s = orc.lz77_encode(data, [4, 1, 5, 0, 3, 24])
    want = serial_tokens(s, 0)
    assert replay(s, want) == data == orc.lz77_decode(s, len(data) + 8)
    for seg_bits, warm in ((512, 128), (2048, 512), (8192, 2048)):
        got, own = stitched_tokens(s, 0, seg_bits, warm)
        assert got == want, (seg_bits,)
    # truncated and damaged streams: same token list either way (whatever it decodes to)
    for t in (s[:-3], s[: len(s) // 2], bytes([s[0] ^ 0x10]) + s[1:], s[:100] + bytes([s[100] ^ 0x41]) + s[101:]):
        assert stitched_tokens(t, 0, 2048, 512)[0] == serial_tokens(t, 0)


def test_raw_offset_bits():
    data = datagen.mixed(20000, 7)
    s = orc.lz77_sa_encode(data, (6, 1, 4, 0, 7, 27, 1))            # rb = 2
    want = serial_tokens(s, 2)
    assert replay(s, want) == data
    assert stitched_tokens(s, 2, 1024, 256)[0] == want
