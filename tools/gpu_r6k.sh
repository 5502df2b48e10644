# Round 6: jobs in flight 10 / 11 / 12 in the steady state, interleaved repeats (r06j: 11 looked 5 % better than 12)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06k}
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 48 --warmup 5 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'single', (d.get('single_job') or {}).get('ms'))" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2 3; do
sw "11 in flight" "X=1" "--pipeline 11" | tee -a gpurun_out/${T}_sweep.txt
sw "10 in flight" "X=1" "--pipeline 10" | tee -a gpurun_out/${T}_sweep.txt
sw "12 in flight" "X=1" "--pipeline 12" | tee -a gpurun_out/${T}_sweep.txt
sw "9 in flight" "X=1" "--pipeline 9" | tee -a gpurun_out/${T}_sweep.txt
done
