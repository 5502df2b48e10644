# Round-3 profile set (run on the GPU box through gpurun): the -m gpu suite, smoke, rocprofv3 kernel traces of the
# headline step, of the context-mixing workload and of dup8_m1, PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE) of the
# context-mixing coder, then the default bench.py line (every workload nested).  Outputs land in gpurun_out/.
R=$GRAFT_REPO_ROOT
T=${1:-r03}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write* gpurun_out/pmc_cm_*
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
CM="--workload cm_m5 --cm-blocks 2048 --cm-block-bytes 65536 --steps 1 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_cm_m5 -o r1 -- $B $CM > /dev/null 2> $R/gpurun_out/rocprof_stats_cm.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch_cm_m5 -o r1 -- $B $CM --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch_cm.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write_cm_m5 -o r1 -- $B $CM --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write_cm.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_cm_a -o r1 -- $B $CM --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_cm_a.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_cm_b -o r1 -- $B $CM --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_cm_b.err
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_dup8_m1 -o r1 -- $B --workload dup8_m1 --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats_dup8.err
cd $R
python profiles/summarize.py gpurun_out $T > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T cm_m5 > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T dup8_m1 > /dev/null 2>&1
python - <<'PY'
import glob, json, sqlite3
out = {}
for d in ("gpurun_out/pmc_cm_a", "gpurun_out/pmc_cm_b"):
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        for name, ctr, n, s in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "cuda" in k or k.startswith("rocprim"):
                continue
            e = out.setdefault(k, {}).setdefault(ctr, {"launches": 0, "sum": 0.0})
            e["launches"] += n; e["sum"] += s
json.dump({"source": "rocprofv3 --pmc (two passes of 8 counters) -- python bench.py --workload cm_m5 --cm-blocks 2048 --cm-block-bytes 65536 --steps 1 --warmup 0 (tools/gpu_profile_r3.sh); sums over the launches",
           "note": "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)", "kernels": out}, open("gpurun_out/r03_pmc_sq_cm_m5.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", {}).get("sum", 0))[:4]:
    g = lambda c: v.get(c, {}).get("sum", 0)
    print(k, {c: g(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT")})
PY
cp profiles/${T}_rocprof_summary*.txt profiles/traffic*.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write* gpurun_out/pmc_cm_a gpurun_out/pmc_cm_b
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -5 gpurun_out/${T}_tests_gpu.log; tail -1 gpurun_out/${T}_smoke.log
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")}, (d.get("cpu_baseline") or {}).get("value"))
for k,v in d.get("workloads",{}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), str(v.get("error",""))[:200], {a:b for a,b in v.items() if a.startswith("verified")}, (v.get("cpu_baseline") or {}).get("value"))
PY
