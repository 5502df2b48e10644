# Round 6: issue priority for the SHA-256 chains of extract (several jobs in flight), interleaved repeats; and jobs in flight 3 / 4 / 5
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06m}
sw() { local out; out=$(env $2 timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'), 'mism', d.get('sha256_mismatches'), {x:k[x] for x in list(k)[:3]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2; do
sw "default" "X=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "chain priority" "ZPQ_SHA256_PRIO=1" "" | tee -a gpurun_out/${T}_sweep.txt
done
sw "5 in flight" "X=1" "--pipeline 5" | tee -a gpurun_out/${T}_sweep.txt
tail -3 gpurun_out/${T}_last.err
