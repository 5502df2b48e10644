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
