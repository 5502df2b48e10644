# Round 6, first GPU call: (1) the driver's command line (--steps 20 --warmup 5) against the 48-step default with the steady-state window;
# (2) a kernel trace of the PRODUCT step at depth 12, fold off and fold on -> profiles/timeline.py; (3) ZPQJ_TIMING phases under load;
# (4) the LDS-slot hypothesis: shallower parse rings + 8 fragment waves/CU, 4 MiB LZ77 segments.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06a}
S0=$(date +%s)
sw() { local out; out=$(env $2 timeout 250 python ${PRE} bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'steps', d['steps'], 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in list(k)[:6]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "driver: steps 20 warmup 5" "X=1" "--steps 20 --warmup 5" | tee gpurun_out/${T}_steady.txt
sw "default: steps 48" "X=1" "" | tee -a gpurun_out/${T}_steady.txt
sw "steps 20 warmup 5, fold ON" "X=1" "--steps 20 --warmup 5 --twins" | tee -a gpurun_out/${T}_steady.txt
echo "[$(( $(date +%s) - S0 )) s] steady"
# (2) traces
for mode in off on; do
  fl=""; [ $mode = on ] && fl="--twins"
  rm -rf /tmp/prof_$mode
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --no-kernel-timing --steps 36 --warmup 5 $fl > /tmp/prof_$mode.json 2> /tmp/prof_$mode.err )
  tail -1 /tmp/prof_$mode.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('traced run, fold $mode:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'))" | tee -a gpurun_out/${T}_steady.txt
  python profiles/timeline.py /tmp/prof_$mode 1500 1500 > gpurun_out/${T}_timeline_product_depth12_fold_$mode.txt 2>&1
  head -12 gpurun_out/${T}_timeline_product_depth12_fold_$mode.txt
  f=$(ls /tmp/prof_$mode/*kernel_stats.csv /tmp/prof_$mode/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -30 $f > gpurun_out/${T}_rocprof_stats_fold_$mode.csv
done
echo "[$(( $(date +%s) - S0 )) s] traces"
# (3) phases of the product call under load (twelve in flight)
ZPQJ_TIMING=1 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --no-kernel-timing --steps 24 --warmup 5 > /dev/null 2> gpurun_out/${T}_phases_under_load.err
grep -i "zpqj\|phase\|ms" gpurun_out/${T}_phases_under_load.err | tail -40 > gpurun_out/${T}_phases_under_load.txt; tail -14 gpurun_out/${T}_phases_under_load.txt
echo "[$(( $(date +%s) - S0 )) s] phases"
# (4) LDS-slot hypothesis
PRE="tools/run_variant.py ring32"
sw "rings 3/2 + 8 fragment waves/CU" "ZPQ_FRAG_WAVES=8" "--steps 24 --warmup 5" | tee gpurun_out/${T}_lds_hypothesis.txt
sw "rings 3/2" "X=1" "--steps 24 --warmup 5" | tee -a gpurun_out/${T}_lds_hypothesis.txt
PRE=""
sw "LZ77 segment 4 MiB" "ZPQ_LZ_SEG=4194304" "--steps 24 --warmup 5" | tee -a gpurun_out/${T}_lds_hypothesis.txt
sw "no kernel timing events" "X=1" "--steps 24 --warmup 5 --no-kernel-timing" | tee -a gpurun_out/${T}_lds_hypothesis.txt
sw "depth 16" "X=1" "--steps 24 --warmup 5 --pipeline 16" | tee -a gpurun_out/${T}_lds_hypothesis.txt
sw "depth 6" "X=1" "--steps 24 --warmup 5 --pipeline 6" | tee -a gpurun_out/${T}_lds_hypothesis.txt
echo "[$(( $(date +%s) - S0 )) s] done"
tail -3 gpurun_out/${T}_last.err
