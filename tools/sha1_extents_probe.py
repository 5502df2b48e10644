"""A/B probe of the fragment SHA-1 pass (zpq_sha1_extents_dev) on the GPU box: lane-per-extent with direct loads
against the LDS-staged form (ZPQ_SHA1_STAGED=1), on fragment-shaped extents (exponential lengths, back to back), plus
a locality diagnostic for the direct form (equal extents: neighbours in a wave adjacent in memory vs scattered).
    python tools/sha1_extents_probe.py            # parent: runs each setting in its own process (env is read once)
"""
import hashlib
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(kind, gib):
    import numpy as np
    import torch
    from zpaqfranz_amd import engine
    eng = engine.Engine(0)
    dev = torch.device("cuda:0")
    total = int(gib * (1 << 30))
    g = torch.Generator(device=dev); g.manual_seed(5)
    data = torch.randint(0, 256, (total,), dtype=torch.uint8, device=dev, generator=g)
    rng = np.random.default_rng(3)
    if kind == "frag":
        lens = np.minimum(4096 + rng.exponential(65536, size=total // 60000).astype(np.int64), 520192)
        lens = lens[np.cumsum(lens) <= total]
        off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    else:
        n = total // 65536
        lens = np.full(n, 65536, dtype=np.int64)
        off = np.arange(n, dtype=np.int64) * 65536
        if kind == "scattered":
            off = off[rng.permutation(n)]
    n = len(lens)
    d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.uint32).view(np.int32)).to(dev)
    dig = torch.zeros(n * 20, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ts = []
    for it in range(4):
        eng.profile(True)
        eng.sha1_extents_dev(data.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, dig.data_ptr())
        eng.sync()
        rep = eng.profile_report()
        eng.profile(False)
        ts.append({k: round(v[1], 3) for k, v in rep.items()})
    h = dig.cpu().numpy().reshape(n, 20)
    pick = list(rng.integers(0, n, size=48)) + [int(np.argmax(lens)), int(np.argmin(lens)), 0, n - 1]
    ok = True
    for i in pick:
        b = bytes(data[int(off[i]):int(off[i]) + int(lens[i])].cpu().numpy().tobytes())
        ok &= hashlib.sha1(b).digest() == h[i].tobytes()
    ms = min(sum(t.values()) for t in ts)
    print("%-10s staged=%s waves=%s order=%s  n=%d  %.1f GB  best %.2f ms  %.0f GB/s  digests ok: %s  %s" % (
        kind, os.environ.get("ZPQ_SHA1_STAGED", "0"), os.environ.get("ZPQ_SHA_WAVES", "2"), "no" if os.environ.get("ZPQ_SHA_NO_ORDER") else "yes",
        n, lens.sum() / 1e9, ms, lens.sum() / 1e6 / ms, ok, ts[-1]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], float(sys.argv[2]))
        sys.exit(0)
    gib = os.environ.get("PROBE_GIB", "12")
    runs = [("frag", {}), ("frag", {"ZPQ_SHA1_STAGED": "1"}), ("frag", {"ZPQ_SHA1_STAGED": "1", "ZPQ_SHA_WAVES": "3"}), ("frag", {"ZPQ_SHA_WAVES": "3"}),
            ("adjacent", {"ZPQ_SHA_NO_ORDER": "1"}), ("scattered", {"ZPQ_SHA_NO_ORDER": "1"}),
            ("adjacent", {"ZPQ_SHA_NO_ORDER": "1", "ZPQ_SHA1_STAGED": "1"}), ("scattered", {"ZPQ_SHA_NO_ORDER": "1", "ZPQ_SHA1_STAGED": "1"})]
    for kind, env in runs:
        e = dict(os.environ); e.update(env)
        t0 = time.time()
        r = subprocess.run([sys.executable, __file__, kind, gib], env=e, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or ("FAILED rc=%d %s" % (r.returncode, r.stderr[-1500:])), "(%.0f s)" % (time.time() - t0), flush=True)
