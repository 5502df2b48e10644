# SQ counters of one bench workload, per kernel: one pass of 8 counters (own run, no trace domains: gpurun refuses --pmc beside them)
# usage: bash tools/gpu_pmc_sq.sh <tag> <workload> [more bench.py arguments]      -> gpurun_out/<tag>_pmc_sq_<workload>.json
R=$GRAFT_REPO_ROOT
export ZPQ_PMC_TAG=$1 ZPQ_PMC_WORK=$2; shift; shift
export ZPQ_PMC_ARGS="$*"
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS -d $R/gpurun_out/pmc_sq -o r1 -- python $R/bench.py --workload $ZPQ_PMC_WORK --no-cpu-baseline --no-verify --steps 1 --warmup 0 "$@" > /dev/null 2> $R/gpurun_out/pmc_sq.err
cd $R
python - <<'PY'
import glob, json, os, sqlite3
out = {}
for f in glob.glob("gpurun_out/pmc_sq/**/*_results.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for name, ctr, n, s in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k:
            continue
        e = out.setdefault(k, {}).setdefault(ctr, {"launches": 0, "sum": 0.0})
        e["launches"] += n; e["sum"] += s
tag, work = os.environ["ZPQ_PMC_TAG"], os.environ["ZPQ_PMC_WORK"]
json.dump({"source": "rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS -- python bench.py --workload %s --steps 1 --warmup 0 %s (tools/gpu_pmc_sq.sh); sums over the launches" % (work, os.environ.get("ZPQ_PMC_ARGS", "")),
           "note": "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)", "kernels": out}, open("gpurun_out/%s_pmc_sq_%s.json" % (tag, work), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", {}).get("sum", 0))[:5]:
    g = lambda c: v.get(c, {}).get("sum", 0)
    print(k, "waves", g("SQ_WAVES"), "valu", g("SQ_INSTS_VALU"), "salu", g("SQ_INSTS_SALU"), "lds", g("SQ_INSTS_LDS"), "active_valu", g("SQ_ACTIVE_INST_VALU"), "busy", g("SQ_BUSY_CYCLES"), "wave_cycles", g("SQ_WAVE_CYCLES"), "wait_any", g("SQ_WAIT_ANY"))
PY
rm -rf gpurun_out/pmc_sq
