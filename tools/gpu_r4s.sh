# Round 4: rehearsal of the driver's default run + hardware-queue A/B of the headline
R=$GRAFT_REPO_ROOT
T=${1:-r04s}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
for q in 16 32 16 32; do
  GPU_MAX_HW_QUEUES=$q ZPQ_BENCH_NO_PLAIN=1 timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 48 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q', d['value'], d['ms_per_step'], d['steps_in_flight'])" | tee -a gpurun_out/${T}_queues.txt
done
echo "[$(( $(date +%s) - S0 )) s] queues"
timeout 900 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "rc=$?"; tail -c 2500 gpurun_out/${T}_bench_default.json; wc -c gpurun_out/${T}_bench_default.json
echo "[$(( $(date +%s) - S0 )) s] done"
