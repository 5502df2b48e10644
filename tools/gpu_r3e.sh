# Round 3: 256 KiB minimum fragmenter segment + wave-per-fragment ids on the folded set; headline with every check;
# kernel trace of the PIPELINED headline (six jobs in flight): are the kernels themselves stretched, or their dispatch?
R=$GRAFT_REPO_ROOT
T=${1:-r03e}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
timeout 300 python -m pytest tests/test_gpu_twins.py tests/test_gpu_parity.py -k "twin or fragmenter or journaling" -x -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests.log; tail -2 gpurun_out/${T}_tests.log; el tests
timeout 300 python bench.py --workload silesia_x256_m1 > gpurun_out/${T}_bench_headline.json 2> gpurun_out/${T}_bench_headline.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${T}_bench_headline.json").read().strip().splitlines()[-1])
    print("headline", d["value"], d["ms_per_step"], "serial", d.get("ms_per_step_serial"), "depth", d.get("steps_in_flight"), {k:v for k,v in d.items() if k.startswith("verified")})
    print(" plain", d.get("every_byte_hashed")); print(" roofline", d.get("roofline")); print(" kernels", d.get("kernels_ms_per_step")); print(" cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("headline failed", e); print(open("gpurun_out/${T}_bench_headline.err").read()[-1500:])
PY
el headline
rm -rf gpurun_out/prof_stats*
cd /tmp; export TMPDIR=/tmp
export ZPQ_BENCH_NO_PLAIN=1
P="python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --no-kernel-timing"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_depth6 -o r1 -- $P --steps 12 --warmup 3 > $R/gpurun_out/${T}_trace_depth6.json 2> $R/gpurun_out/rocprof_stats6.err
cd $R; python profiles/summarize.py gpurun_out $T depth6 > /dev/null 2>&1; cp profiles/${T}_rocprof_summary_depth6.txt gpurun_out/ 2>/dev/null; head -22 gpurun_out/${T}_rocprof_summary_depth6.txt; tail -1 gpurun_out/${T}_trace_depth6.json | cut -c1-300; el trace6
rm -rf gpurun_out/prof_stats*
el done
