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
