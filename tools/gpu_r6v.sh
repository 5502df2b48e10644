# Round 6: text_m2 jobs in flight 3 / 4 / 5, interleaved
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06v}
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload text_m2 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'))" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2; do
sw "3 in flight" "X=1" "--steps 12" | tee -a gpurun_out/${T}_sweep.txt
sw "4 in flight" "X=1" "--steps 12 --pipeline 4" | tee -a gpurun_out/${T}_sweep.txt
sw "5 in flight" "X=1" "--steps 12 --pipeline 5" | tee -a gpurun_out/${T}_sweep.txt
done
tail -2 gpurun_out/${T}_last.err
