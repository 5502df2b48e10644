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


def replay_in_steps(s, toks, step=64, short=32):
    """Model of lz77_copy_kernel: `step` tokens at a time, output positions by prefix sum; short tokens are copied in
    rounds -- a token is ready when it is a literal, when its source ends before the output of the first token still
    pending, or when it is that first token itself; a long token is copied alone and splits the step.  Every byte a
    ready token reads must already be written (asserted): that is what makes the rounds equal to the serial replay."""
    total = sum(ln for _, ln, _ in toks)
    out = bytearray(total)
    written = bytearray(total)                               # 1 where out[] holds its final byte
    rounds = 0

    def put(pos, data):
        out[pos:pos + len(data)] = data
        written[pos:pos + len(data)] = b"\x01" * len(data)

    def lit_bytes(x, ln):
        b0, sh = x >> 3, x & 7
        return bytes((((s[b0 + j] | ((s[b0 + j + 1] << 8) if sh else 0)) >> sh) & 255) for j in range(ln))
    op = 0
    for t0 in range(0, len(toks), step):
        win = toks[t0:t0 + step]
        pos = [op]
        for _, ln, _ in win:
            pos.append(pos[-1] + ln)
        cur = 0
        while cur < len(win):
            L = next((j for j in range(cur, len(win)) if win[j][1] > short), len(win))
            pend = list(range(cur, L))
            while pend:                                      # rounds over the short tokens [cur, L)
                f = pend[0]
                ready = [j for j in pend if win[j][0] == "lit" or j == f or pos[j] - win[j][2] + win[j][1] <= pos[f]]
                for j in ready:                              # all of these run at once on the GPU: only read what is written
                    kind, ln, x = win[j]
                    if kind == "lit":
                        continue
                    src = pos[j] - x
                    assert x >= 1 and src >= 0
                    if j != f or x >= ln:
                        assert all(written[src:src + ln]), (t0, j)
                for j in ready:
                    kind, ln, x = win[j]
                    if kind == "lit":
                        put(pos[j], lit_bytes(x, ln))
                    elif x < ln:                             # self-overlapping: byte by byte, it is the first pending token
                        assert j == f
                        for q in range(ln):
                            put(pos[j] + q, bytes([out[pos[j] + q - x]]))
                    else:
                        put(pos[j], bytes(out[pos[j] - x:pos[j] - x + ln]))
                pend = [j for j in pend if j not in ready]
                rounds += 1
            if L < len(win):                                 # one long token, alone
                kind, ln, x = win[L]
                if kind == "lit":
                    put(pos[L], lit_bytes(x, ln))
                else:
                    for q in range(ln):
                        put(pos[L] + q, bytes([out[pos[L] + q - x]]))
            cur = L + 1
        op = pos[-1]
    return bytes(out), rounds


@pytest.mark.parametrize("seed,kind", [(11, "mixed"), (12, "text"), (13, "binary")])
def test_replay_in_rounds_equals_serial_replay(seed, kind):
    data = {"mixed": datagen.mixed, "text": datagen.text_like, "binary": datagen.binary_like}[kind](30000, seed)
    data = data + data[100:9000] + bytes(700) + b"ab" * 400 + data[:50]        # long matches, a run (self-overlap), short periods
    s = orc.lz77_encode(data, [4, 1, 5, 0, 3, 24])
    toks = serial_tokens(s, 0)
    got, rounds = replay_in_steps(s, toks)
    assert got == data == replay(s, toks)
    assert rounds < len(toks) // 4                           # rounds, not one pass per token (about 9 per step of 64 on this data)
