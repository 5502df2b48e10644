"""The one-pass form of the LZ77 level-1 decision (lz77_waves.inc): LZBuffer scores a candidate with
score = 8 l - lg(offset) - 2 (lit > 0) - 11 (ZSFX/libzpaq.cpp:6396-6408), and the GPU needs the decision for both values of
(lit > 0).  The two runs accept the same candidates unless one is accepted with a score of 1 or 2; the kernel therefore
makes one run and repeats it only where that happened.  Here: a model of both forms over random candidate lists in which
whether a candidate passes its byte test is an ARBITRARY function of (candidate, current best length) -- more than the
data can do -- and scores crowd around the thresholds."""
import random


def two_runs(cands, ok, mm):
    out = []
    for f in (0, 1):
        blen, bp, bscore = mm - 1, 0, 0
        for k, (l, p, lgo) in enumerate(cands):
            if blen < 128 and p is not None and ok(k, blen):
                score = l * 8 - lgo - 2 * f - 11
                if score > bscore:
                    blen, bp, bscore = l, p, score
        out.append((blen, bp) if (bp != 0 and bscore > 0 and blen >= mm) else (0, 0))
    return out


def one_run(cands, ok, mm):
    blen, bp, bscore, thin = mm - 1, 0, 0, False
    for k, (l, p, lgo) in enumerate(cands):
        if blen < 128 and p is not None and ok(k, blen):
            score = l * 8 - lgo - 11
            if score > bscore:
                thin = thin or score <= 2
                blen, bp, bscore = l, p, score
    r0 = (blen, bp) if (bp != 0 and bscore > 0 and blen >= mm) else (0, 0)
    if not thin:
        r1 = (blen, bp) if (bp != 0 and bscore > 2 and blen >= mm) else (0, 0)
        return [r0, r1], False
    blen, bp, bscore = mm - 1, 0, 0
    for k, (l, p, lgo) in enumerate(cands):
        if blen < 128 and p is not None and ok(k, blen):
            score = l * 8 - lgo - 2 - 11
            if score > bscore:
                blen, bp, bscore = l, p, score
    return [r0, (blen, bp) if (bp != 0 and bscore > 0 and blen >= mm) else (0, 0)], True


def test_one_run_with_a_rare_second_equals_two_runs():
    rng = random.Random(5)
    thin_seen = differing = 0
    for _ in range(60000):
        mm = rng.choice((4, 5, 6))
        cands = []
        for _k in range(8):
            if rng.random() < 0.3:
                cands.append((0, None, 0))
            else:
                l = rng.choice((3, 4, 4, 5, 5, 6, 7, 12, 32, 130, 200))
                cands.append((l, rng.randrange(1, 1 << 24), rng.randrange(1, 25)))
        table = {(k, b): rng.random() < 0.7 for k in range(8) for b in (3, 4, 5, 6, 7, 12, 32, 130, 200)}
        ok = lambda k, b: table[(k, b)]
        want = two_runs(cands, ok, mm)
        got, thin = one_run(cands, ok, mm)
        assert got == want, (cands, mm, got, want)
        thin_seen += thin
        differing += want[0] != want[1]
    assert thin_seen > 500 and differing > 100          # the rare path and decisions that depend on (lit > 0) were both exercised


def test_one_bit_per_xor_byte_is_all_the_evaluator_needs():
    """The four-wave kernel's evaluator (lz77_waves.inc, LEAN) keeps of a candidate's 32 XOR bytes one bit each: bit 8b + w of M
    is set when byte b of word w differs.  From M alone: the length (first byte that differs, in byte order) and the reference's
    in[p+blen-1] == in[i+blen-1] for any blen - 1 < 32.  The arithmetic, restated: bit 7 of every byte of
    ((x & 0x7f7f7f7f) + 0x7f7f7f7f) | x is set exactly when that byte of x is not zero (no carry leaves a byte)."""
    rng = random.Random(11)

    def nz(x):
        return ((((x & 0x7f7f7f7f) + 0x7f7f7f7f) & 0xffffffff) | x) & 0xffffffff

    for _ in range(20000):
        # XOR of two 32-byte strings that agree on a random prefix and here and there behind it
        same = rng.choice((0, 1, 3, 4, 7, 8, 15, 16, 17, 31, 32))
        xor = bytes(0 if (i < same or rng.random() < 0.4) else rng.randrange(1, 256) for i in range(32))
        words = [int.from_bytes(xor[4 * w:4 * w + 4], "little") for w in range(8)]
        for w in range(8):
            flags = nz(words[w])
            assert [(flags >> (8 * b + 7)) & 1 for b in range(4)] == [int(xor[4 * w + b] != 0) for b in range(4)]
        M = 0
        for w in range(8):
            M |= (nz(words[w]) >> (7 - w)) & (0x01010101 << w)
        length = 32
        for b in range(4):
            byte = (M >> (8 * b)) & 255
            first = 4 * (((byte | 0x100) & -(byte | 0x100)).bit_length() - 1) + b          # 4 * ctz(byte | 0x100) + b
            length = min(length, first)
        want = next((i for i in range(32) if xor[i]), 32)
        assert length == want, (xor.hex(), length, want)
        for idx in range(32):
            w = idx >> 2
            eqb = ((M >> (((idx & 3) << 3) | (w & 7))) & 1) == 0
            assert eqb == (xor[idx] == 0)
