# Round 6: is one resume wave per CU (instead of two) a real gain?  interleaved repeats
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06j}
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 48 --warmup 5 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'single', (d.get('single_job') or {}).get('ms'))" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2 3; do
sw "default" "X=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "resume waves 1/CU" "ZPQ_FRAG_RESUME_WAVES=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "resume waves 1/CU, 11 in flight" "ZPQ_FRAG_RESUME_WAVES=1" "--pipeline 11" | tee -a gpurun_out/${T}_sweep.txt
done
