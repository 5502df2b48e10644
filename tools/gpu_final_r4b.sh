# Round-4 closing run, part 2: the default bench.py line (every workload nested), then kernel traces of the serial headline step
# (with FETCH / WRITE passes) and of the many-block add
R=$GRAFT_REPO_ROOT
T=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
timeout 1000 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d.get("ms_per_step_serial"), {k:v for k,v in d.items() if k.startswith("verified")}, (d.get("cpu_baseline") or {}).get("value"), d.get("every_byte_hashed"))
print(json.dumps(d.get("workloads_summary")))
PY
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- $B --workload silesia_x256_m1 --pipeline 1 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write.err
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_dup8_m1 -o r1 -- $B --workload dup8_m1 --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats_dup8.err
cd $R
python profiles/summarize.py gpurun_out $T > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T dup8_m1 > /dev/null 2>&1
cp profiles/${T}_rocprof_summary*.txt profiles/traffic.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
head -16 gpurun_out/${T}_rocprof_summary.txt; head -8 gpurun_out/${T}_rocprof_summary_dup8_m1.txt
