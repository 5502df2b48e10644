"""Deterministic synthetic inputs shared by the parity tests and bench.py (no corpus is available
offline).  Everything is seeded numpy; the same bytes are produced here and on the GPU box."""
import numpy as np

_WORDS = None


def _words(rng):
    global _WORDS
    if _WORDS is None:
        r = np.random.default_rng(12345)
        alpha = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
        p = p / p.sum()
        _WORDS = [bytes(r.choice(alpha, size=int(r.integers(1, 10)), p=p)) for _ in range(4096)]
    return _WORDS


def text_like(n, seed):
    """Zipf-distributed words, punctuation, occasional capitals and digits."""
    rng = np.random.default_rng(seed)
    w = _words(rng)
    idx = np.minimum(rng.zipf(1.3, size=n // 4 + 16) - 1, len(w) - 1)
    parts = []
    total = 0
    for k, i in enumerate(idx):
        s = w[i]
        r = k * 2654435761 % 97
        if r == 0:
            s = s.capitalize()
        if r == 1:
            s = s + b"."
        if r == 2:
            s = s + b","
        if r == 3:
            s = b"%d" % (k % 1000)
        if r == 4:
            s = s + b"\r\n"
        parts.append(s)
        total += len(s) + 1
        if total >= n:
            break
    return b" ".join(parts)[:n].ljust(n, b" ")


def binary_like(n, seed):
    """Structured binary: records with small-integer fields, repeated tables, zero runs."""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=np.uint8)
    pos = 0
    table = rng.integers(0, 256, size=4096, dtype=np.uint8)
    while pos < n:
        kind = rng.integers(0, 5)
        ln = int(rng.integers(16, 2048))
        ln = min(ln, n - pos)
        if kind == 0:
            out[pos:pos + ln] = 0
        elif kind == 1:
            s = int(rng.integers(0, 4096 - 1))
            seg = np.resize(table[s:s + max(1, min(ln, 4096 - s))], ln)
            out[pos:pos + ln] = seg
        elif kind == 2:
            out[pos:pos + ln] = rng.integers(0, 256, size=ln, dtype=np.uint8)
        elif kind == 3:
            rec = rng.integers(0, 16, size=ln, dtype=np.uint8)
            rec[::4] = 0xE8 if rng.integers(0, 2) else 0x8B
            out[pos:pos + ln] = rec
        else:
            back = int(rng.integers(1, pos + 1)) if pos else 0
            if back:
                src = out[pos - back:pos - back + ln]
                out[pos:pos + len(src)] = src
                ln = max(1, len(src))
            else:
                out[pos:pos + ln] = 7
        pos += ln
    return out.tobytes()


def mixed(n, seed):
    """Concatenation of text-like and binary-like stretches (a stand-in for a Silesia member)."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    parts, total, k = [], 0, 0
    while total < n:
        ln = int(min(n - total, rng.integers(1 << 12, 1 << 17)))
        parts.append(text_like(ln, seed * 1000 + k) if (k + seed) % 3 else binary_like(ln, seed * 1000 + k))
        total += ln
        k += 1
    return b"".join(parts)[:n]


def random_bytes(n, seed):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


# ---------------------------------------------------------------------------------------------------
# Silesia-shaped synthetic corpus (the real corpus cannot be fetched offline).  Member names and
# sizes are Silesia's (211 938 580 bytes in 12 files); contents are seeded synthetic material of a
# similar character (natural-language-like text, tagged text, structured binary, noisy sensor-like
# binary) with long-range repeats, so that fragmenting, dedup and LZ77 see realistic statistics.
# ---------------------------------------------------------------------------------------------------
SILESIA = [("dickens", 10192446, "text"), ("mozilla", 51220480, "exe"), ("mr", 9970564, "sensor"),
           ("nci", 33553445, "table"), ("ooffice", 6152192, "exe"), ("osdb", 10085684, "records"),
           ("reymont", 6627202, "text"), ("samba", 21606400, "mixed"), ("sao", 7251944, "sensor"),
           ("webster", 41458703, "text"), ("x-ray", 8474240, "sensor"), ("xml", 5345280, "tagged")]


def _vocab(rng, nwords, alphabet, lo, hi, sep):
    lens = rng.integers(lo, hi + 1, size=nwords)
    total = int(lens.sum() + nwords)
    buf = alphabet[rng.integers(0, len(alphabet), size=total)].copy()
    off = np.concatenate(([0], np.cumsum(lens + 1)[:-1]))
    buf[off + lens] = sep
    return buf, off, lens + 1


def _from_vocab(rng, n, buf, off, lens, zipf_a):
    m = int(n // max(2.0, lens.mean()) + 64)
    ids = np.minimum(rng.zipf(zipf_a, size=m) - 1, len(off) - 1)
    L = lens[ids]
    ends = np.cumsum(L)
    starts = ends - L
    idx = np.repeat(off[ids] - starts, L) + np.arange(int(ends[-1]))
    out = buf[idx]
    while len(out) < n:
        out = np.concatenate((out, out))
    return out[:n]


def _fresh(kind, n, rng):
    letters = np.frombuffer(b"eeeeeeetttttaaaaooooiiiinnnnsssshhhrrrddlllcuumwfgypbvk", dtype=np.uint8)
    if kind == "text":
        v = _vocab(rng, 30000, letters, 1, 11, 32)
        out = _from_vocab(rng, n, *v, 1.25)
        out[rng.integers(0, n, size=n // 90)] = 46      # full stops
        out[rng.integers(0, n, size=n // 400)] = 10     # newlines
        return out
    if kind == "tagged":
        v = _vocab(rng, 400, np.frombuffer(b"abcdefghijklmnopqrstuvwxyz<>/=\"_", dtype=np.uint8), 3, 18, 62)
        return _from_vocab(rng, n, *v, 1.15)
    if kind == "table":
        v = _vocab(rng, 300, np.frombuffer(b"0123456789 .-", dtype=np.uint8), 4, 30, 10)
        return _from_vocab(rng, n, *v, 1.1)
    if kind == "records":
        v = _vocab(rng, 20000, np.arange(256, dtype=np.uint8), 6, 40, 0)
        return _from_vocab(rng, n, *v, 1.4)
    if kind == "sensor":
        walk = np.cumsum(rng.integers(-3, 4, size=n // 2 + 1, dtype=np.int32)).astype(np.uint16)
        out = np.empty(n + 2, dtype=np.uint8)
        out[0:2 * len(walk):2] = (walk & 255).astype(np.uint8)[: len(out[0::2])]
        out[1:2 * len(walk):2] = (walk >> 8).astype(np.uint8)[: len(out[1::2])]
        return out[:n]
    if kind == "exe":
        ops = np.frombuffer(bytes([0x8B, 0x89, 0xE8, 0xFF, 0x0F, 0x83, 0x48, 0x00, 0x00, 0x24, 0x45, 0x74, 0x75, 0xC3, 0x55]), dtype=np.uint8)
        v = _vocab(rng, 50000, ops, 2, 9, 0x90)
        out = _from_vocab(rng, n, *v, 1.3)
        k = n // 6
        out[rng.integers(0, n, size=k)] = rng.integers(0, 256, size=k, dtype=np.uint8)
        return out
    # mixed
    parts, tot, i = [], 0, 0
    kinds = ["text", "exe", "tagged", "records"]
    while tot < n:
        ln = int(min(n - tot, rng.integers(1 << 16, 1 << 21)))
        parts.append(_fresh(kinds[i % 4], ln, rng))
        tot += ln
        i += 1
    return np.concatenate(parts)[:n]


def silesia_member(name, n, kind, seed):
    rng = np.random.default_rng([seed, sum(name.encode())])
    out = _fresh(kind, n, rng)
    # long-range repeats: copy earlier spans forward (the redundancy LZ77 and dedup feed on)
    k = n // 2500
    dst = np.sort(rng.integers(64, max(65, n - 600), size=k))
    ln = rng.integers(12, 500, size=k)
    back = (2.0 ** rng.uniform(6, 22, size=k)).astype(np.int64)
    for d, l, b in zip(dst.tolist(), ln.tolist(), back.tolist()):
        s = d - b
        if s >= 0 and d + l <= n:
            out[d:d + l] = out[s:s + l]
    return out.tobytes()


def silesia_like(seed=0, scale=1.0):
    """List of (name, bytes); scale < 1 shrinks every member proportionally (for tests)."""
    return [(nm, silesia_member(nm, max(1, int(sz * scale)), kind, seed)) for nm, sz, kind in SILESIA]
