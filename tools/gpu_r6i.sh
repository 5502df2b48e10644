# Round 6: headline knobs again, now that the window is the steady state (run-to-run spread of the default: 87-101 ms)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06i}
S0=$(date +%s)
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 36 --warmup 5 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'), {x:k[x] for x in list(k)[:5]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "default" "X=1" "" | tee gpurun_out/${T}_sweep.txt
sw "default again" "X=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "resume waves 1/CU" "ZPQ_FRAG_RESUME_WAVES=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "resume waves 4/CU" "ZPQ_FRAG_RESUME_WAVES=4" "" | tee -a gpurun_out/${T}_sweep.txt
sw "fragment waves 5/CU" "ZPQ_FRAG_WAVES=5" "" | tee -a gpurun_out/${T}_sweep.txt
sw "fragment waves 7/CU" "ZPQ_FRAG_WAVES=7" "" | tee -a gpurun_out/${T}_sweep.txt
sw "fragment waves 8/CU" "ZPQ_FRAG_WAVES=8" "" | tee -a gpurun_out/${T}_sweep.txt
sw "crossing budget 128 KiB" "ZPQ_FRAG_BUDGET=131072" "" | tee -a gpurun_out/${T}_sweep.txt
sw "crossing budget 512 KiB" "ZPQ_FRAG_BUDGET=524288" "" | tee -a gpurun_out/${T}_sweep.txt
sw "10 in flight" "X=1" "--pipeline 10" | tee -a gpurun_out/${T}_sweep.txt
sw "11 in flight" "X=1" "--pipeline 11" | tee -a gpurun_out/${T}_sweep.txt
sw "GPU_MAX_HW_QUEUES 8" "GPU_MAX_HW_QUEUES=8" "" | tee -a gpurun_out/${T}_sweep.txt
sw "default third" "X=1" "" | tee -a gpurun_out/${T}_sweep.txt
echo "[$(( $(date +%s) - S0 )) s] done"
