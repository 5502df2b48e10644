#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of one bench.py run into the text summaries kept under
profiles/.  Usage: python profiles/summarize.py gpurun_out rNN

  gpurun_out/prof_stats/*_results.db   rocprofv3 --kernel-trace --stats -- python bench.py ...
  gpurun_out/prof_fetch/*_results.db   rocprofv3 --pmc FETCH_SIZE     -- python bench.py --steps 1 --warmup 0 ...
  gpurun_out/prof_write/*_results.db   rocprofv3 --pmc WRITE_SIZE     -- ...

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read; the run itself carries a calibration point (the
rocclr copyBuffer dispatches that replicate the 211 938 580-byte corpus: FETCH_SIZE = 103 498 KiB =
0.50 x bytes copied), so read bytes below are reported raw and x2."""
import glob
import os
import sqlite3
import sys


def q(dbdir):
    f = glob.glob(os.path.join(dbdir, "*_results.db"))
    return sqlite3.connect(f[0]).cursor() if f else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main(root, tag, work=""):
    """work = "" for the default workload (directories prof_stats/fetch/write, files <tag>_rocprof_summary.txt and
    traffic.json); otherwise a label: directories prof_stats_<work>..., file <tag>_rocprof_summary_<work>.txt"""
    out = []
    sfx = "_" + work if work else ""
    cur = q(os.path.join(root, "prof_stats" + sfx))
    if cur:
        out.append("# rocprofv3 --kernel-trace --stats -- python bench.py %s  (%s)\n" % ("--workload " + work if work else "--steps 3 --warmup 1", tag))
        out.append("%-34s %6s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            out.append("%-34s %6d %14.1f %14.1f %7.2f" % (short(name)[:34], calls, total / 1e3, avg / 1e3, pct))
        out.append("")
    for sub, ctr in (("prof_fetch" + sfx, "FETCH_SIZE"), ("prof_write" + sfx, "WRITE_SIZE")):
        cur = q(os.path.join(root, sub))
        if not cur:
            continue
        out.append("# rocprofv3 --pmc %s -- python bench.py --steps 1 --warmup 0   (KiB per dispatch, summed per kernel)" % ctr)
        out.append("%-34s %6s %16s %16s" % ("kernel", "calls", "sum_KiB", "avg_KiB"))
        rows = cur.execute("select kernel_name, count(*), sum(value), avg(value) from counters_collection where counter_name=? group by kernel_name order by sum(value) desc", (ctr,))
        for name, calls, sm, av in rows:
            out.append("%-34s %6d %16.1f %16.1f" % (short(name)[:34], calls, sm, av))
        out.append("")
    # bytes per launch for bench.py's roofline.traffic: FETCH_SIZE*2 (gfx950 correction) + WRITE_SIZE
    per, tot, launches = {}, {}, {}
    for sub, ctr, mul in (("prof_fetch" + sfx, "FETCH_SIZE", 2.0), ("prof_write" + sfx, "WRITE_SIZE", 1.0)):
        cur = q(os.path.join(root, sub))
        if not cur:
            continue
        for name, calls, sm in cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
            k = short(name)
            k = "fragment_resume_kernel" if k.startswith("fragment_spec_kernel<true") else k.split("<")[0]   # names bench.py uses
            k = {"lz77_spec3_kernel": "lz77_spec_kernel", "lz77_direct3_kernel": "lz77_direct_kernel", "lz77_direct4_kernel": "lz77_direct_kernel"}.get(k, k)
            if k.startswith("rocprim::"):
                k = "rocprim"                 # (sort and scan passes under the suffix array: bench.py's scope sa_radix_sort_pairs)
            per[k] = per.get(k, 0) + int(sm * 1024 * mul / max(1, calls))
            tot[k] = tot.get(k, 0) + int(sm * 1024 * mul)
            launches[k] = max(launches.get(k, 0), calls)
    if per:
        import json
        # whole jobs the counter passes ran (bench.py runs sizing and verification steps beside the timed one): a kernel launched once per job
        marker = {"text_m2": "sha1_chain_kernel", "extract_m1": "unframe_walk_kernel"}.get(work, "dedup_insert_kernel")
        json.dump({"source": "profiles/%s_rocprof_summary%s.txt: per launch, FETCH_SIZE*2 (gfx950 correction, calibrated on the copyBuffer "
                             "dispatches of the same run) + WRITE_SIZE" % (tag, sfx),
                   "jobs_profiled": launches.get(marker, 0), "jobs_marker_kernel": marker,
                   "bytes_per_launch": dict(sorted(per.items(), key=lambda kv: -kv[1])),
                   "bytes_total": dict(sorted(tot.items(), key=lambda kv: -kv[1])), "launches": launches},
                  open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic%s.json" % sfx), "w"), indent=1)
    txt = "\n".join(out)
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "%s_rocprof_summary%s.txt" % (tag, sfx)), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
