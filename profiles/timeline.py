#!/usr/bin/env python3
"""What the chip does while several jobs are in flight: from the rocpd database of a `rocprofv3 --kernel-trace` run,
the dispatches' start / end times and launch shapes give, over the traced interval,
  * the time no kernel runs at all (the host, not the GPU, is what the step waits for),
  * the time only few-wave kernels run (serial chains: block checksums, segment parses, crossing walks),
  * the time chip-filling kernels (>= 1024 waves: one per SIMD) run, and how many of them at once,
  * per kernel: launches, summed duration, its share of the 'wide' time, mean number of waves.
Usage: python profiles/timeline.py <dir with *_results.db> [last_ms [skip_end_ms]]   (the window analysed: `last_ms` of the trace
ending `skip_end_ms` before its last dispatch ends -- the steady state of a pipelined run without warm-up or drain)"""
import glob
import os
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:40]


def load(dbdir):
    f = glob.glob(os.path.join(dbdir, "*_results.db")) + glob.glob(os.path.join(dbdir, "*", "*_results.db"))
    if not f:
        raise SystemExit("no *_results.db under " + dbdir)
    con = sqlite3.connect(f[0])
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "kernels" in names:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        want = {}
        for c in cols:
            lc = c.lower()
            if lc in ("name", "kernel_name"): want["name"] = c
            elif lc == "start": want["start"] = c
            elif lc == "end": want["end"] = c
            elif lc in ("grid_x", "grid_size_x"): want["gx"] = c
            elif lc in ("grid_y", "grid_size_y"): want["gy"] = c
            elif lc in ("grid_z", "grid_size_z"): want["gz"] = c
            elif lc in ("workgroup_x", "workgroup_size_x"): want["wx"] = c
            elif lc in ("workgroup_y", "workgroup_size_y"): want["wy"] = c
            elif lc in ("workgroup_z", "workgroup_size_z"): want["wz"] = c
        if all(k in want for k in ("name", "start", "end", "gx", "wx")):
            sel = "select %s,%s,%s,%s,%s,%s,%s,%s,%s from kernels" % tuple(want.get(k, "1") for k in ("name", "start", "end", "gx", "gy", "gz", "wx", "wy", "wz"))
            return list(cur.execute(sel))
    sys.stderr.write("tables/views: %s\n" % names)
    for n in names:
        if "dispatch" in n.lower():
            sys.stderr.write("%s: %s\n" % (n, [r[1] for r in cur.execute("pragma table_info(%s)" % n)]))
    raise SystemExit("no usable 'kernels' view in the database")


def main(dbdir, last_ms=None, skip_end_ms=0.0):
    rows = load(dbdir)
    ev = []
    for name, s, e, gx, gy, gz, wx, wy, wz in rows:
        wg = max(1, wx) * max(1, wy or 1) * max(1, wz or 1)
        threads = max(1, gx) * max(1, gy or 1) * max(1, gz or 1)        # HIP grid sizes are in threads here
        waves = (threads // wg) * -(-wg // 64)
        nm = short(name)
        if nm == "sha1_chain_kernel":            # one symbol, two launch shapes: a wave per 16 MiB block (13 waves, the block checksums) / per fragment (thousands)
            nm += " [block checksums]" if waves <= 64 else " [fragment ids]"
        ev.append((int(s), int(e), nm, max(1, waves)))
    t_lo = min(s for s, _, _, _ in ev); t_hi = max(e for _, e, _, _ in ev)
    b = t_hi - skip_end_ms * 1e6
    a = t_lo if last_ms is None else max(t_lo, b - last_ms * 1e6)
    ev = [(max(s, a), min(e, b), n, w) for s, e, n, w in ev if e > a and s < b]
    pts = []
    for i, (s, e, n, w) in enumerate(ev):
        pts.append((s, 1, i)); pts.append((e, -1, i))
    pts.sort()
    live = set()
    idle = narrow = wide = 0.0
    wide_hist = {}
    share = {}                      # kernel -> time-weighted share of the chip while it ran (waves / 1024, capped, split among the wide ones)
    last = a
    for t, kind, i in pts:
        dt = t - last
        if dt > 0:
            if not live:
                idle += dt
            else:
                tot = sum(min(1024, ev[j][3]) for j in live)
                nwide = sum(1 for j in live if ev[j][3] >= 1024)
                if nwide:
                    wide += dt; wide_hist[nwide] = wide_hist.get(nwide, 0.0) + dt
                else:
                    narrow += dt
                for j in live:
                    share[ev[j][2]] = share.get(ev[j][2], 0.0) + dt * min(1024, ev[j][3]) / max(1024, tot)
        last = t
        if kind == 1: live.add(i)
        else: live.discard(i)
    span = b - a
    print("# timeline of %s  (%.1f ms traced, %d dispatches)" % (dbdir, span / 1e6, len(ev)))
    print("no kernel running      %8.1f ms  %5.1f %%" % (idle / 1e6, 100 * idle / span))
    print("only few-wave kernels  %8.1f ms  %5.1f %%" % (narrow / 1e6, 100 * narrow / span))
    print("chip-filling kernel(s) %8.1f ms  %5.1f %%   by number at once: %s" % (wide / 1e6, 100 * wide / span,
          {k: round(v / 1e6, 1) for k, v in sorted(wide_hist.items())}))
    per = {}
    for s, e, n, w in ev:
        c, d, ww = per.get(n, (0, 0.0, 0.0))
        per[n] = (c + 1, d + (e - s), ww + w)
    print("\n%-40s %6s %10s %9s %12s %10s" % ("kernel", "calls", "sum_ms", "waves", "chip-ms", "avg_ms"))
    print("# chip-ms: time x (waves / 1024, capped at 1, shared with whatever else runs): SIMD-seats the kernel held, in ms of the whole chip")
    for n, (c, d, ww) in sorted(per.items(), key=lambda kv: -share.get(kv[0], 0.0))[:28]:
        print("%-40s %6d %10.1f %9d %12.1f %10.2f" % (n, c, d / 1e6, ww / c, share.get(n, 0.0) / 1e6, d / 1e6 / c))
    print("%-40s %6s %10s %9s %12.1f  (= %.1f %% of the traced time)" % ("total", "", "", "", sum(share.values()) / 1e6, 100 * sum(share.values()) / span))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else None, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
