# Round 5, tenth GPU call: jobs in flight (depth) with the round's kernels, every byte hashed.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1
T=${1:-r05j}
S0=$(date +%s)
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 48 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in list(k)[:5]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "12 in flight (default)" "X=1" "" | tee gpurun_out/${T}_sweep_depth.txt
sw "8 in flight" "X=1" "--pipeline 8" | tee -a gpurun_out/${T}_sweep_depth.txt
sw "10 in flight" "X=1" "--pipeline 10" | tee -a gpurun_out/${T}_sweep_depth.txt
sw "14 in flight" "X=1" "--pipeline 14" | tee -a gpurun_out/${T}_sweep_depth.txt
sw "12 in flight, 7 fragment waves/CU" "ZPQ_FRAG_WAVES=7" "" | tee -a gpurun_out/${T}_sweep_depth.txt
sw "12 in flight, 5 fragment waves/CU" "ZPQ_FRAG_WAVES=5" "" | tee -a gpurun_out/${T}_sweep_depth.txt
echo "[$(( $(date +%s) - S0 )) s] done"
tail -3 gpurun_out/${T}_last.err
