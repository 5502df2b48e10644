# Round-3 closing run on the GPU box: the default bench.py line (every workload nested), the -m gpu suite, smoke,
# then kernel traces of the headline step and of the context-mixing workload and the FETCH/WRITE passes of the
# headline step.  Outputs land in gpurun_out/.
R=$GRAFT_REPO_ROOT
T=${1:-r03b}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
timeout 1000 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")}, (d.get("cpu_baseline") or {}).get("value"))
for k,v in d.get("workloads",{}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), str(v.get("error",""))[:200], {a:b for a,b in v.items() if a.startswith("verified")}, (v.get("cpu_baseline") or {}).get("value"))
PY
timeout 700 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
tail -4 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
tail -1 gpurun_out/${T}_smoke.log
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
CM="--workload cm_m5 --cm-blocks 2048 --cm-block-bytes 65536 --steps 1 --warmup 1"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_cm_m5 -o r1 -- $B $CM > /dev/null 2> $R/gpurun_out/rocprof_stats_cm.err
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write.err
cd $R
python profiles/summarize.py gpurun_out $T > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T cm_m5 > /dev/null 2>&1
cp profiles/${T}_rocprof_summary*.txt profiles/traffic.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
head -12 gpurun_out/${T}_rocprof_summary.txt
