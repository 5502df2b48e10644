# SQ counters of the many-block add (config 4 at 1-GPU size: 1091 blocks, a workgroup of four waves each): one pass of 8 counters, per kernel
# usage: bash tools/gpu_pmc_dup8.sh [output tag, default r05]
R=$GRAFT_REPO_ROOT
export ZPQ_PMC_TAG=${1:-r05}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_dup8
timeout 280 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS -d $R/gpurun_out/pmc_dup8 -o r1 -- python $R/bench.py --workload dup8_m1 --no-cpu-baseline --no-verify --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_dup8.err
cd $R
python - <<'PY'
import glob, json, sqlite3
out = {}
for f in glob.glob("gpurun_out/pmc_dup8/**/*_results.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for name, ctr, n, s in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k:
            continue
        e = out.setdefault(k, {}).setdefault(ctr, {"launches": 0, "sum": 0.0})
        e["launches"] += n; e["sum"] += s
json.dump({"source": "rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS -- python bench.py --workload dup8_m1 --steps 1 --warmup 0 (tools/gpu_pmc_dup8.sh); sums over the launches",
           "note": "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)", "kernels": out}, open("gpurun_out/%s_pmc_sq_dup8.json" % __import__("os").environ.get("ZPQ_PMC_TAG", "r05"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", {}).get("sum", 0))[:5]:
    g = lambda c: v.get(c, {}).get("sum", 0)
    print(k, "waves", g("SQ_WAVES"), "valu", g("SQ_INSTS_VALU"), "salu", g("SQ_INSTS_SALU"), "lds", g("SQ_INSTS_LDS"), "active_valu", g("SQ_ACTIVE_INST_VALU"), "busy", g("SQ_BUSY_CYCLES"), "wave_cycles", g("SQ_WAVE_CYCLES"), "wait_any", g("SQ_WAIT_ANY"))
PY
rm -rf gpurun_out/pmc_dup8
