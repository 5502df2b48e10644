# Round 6, sixth GPU call: a communicator per add in flight (one rank through the multi-rank product path, C function pointers, no turn
# order), the two-rank gloo tests again, extract_m1 with twelve timed jobs: lanes per chain 16 / 8 / 4 and three jobs in flight.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06f}
S0=$(date +%s)
for i in 1 2; do
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 2> gpurun_out/${T}_bench_rccl1.err | tail -1 > gpurun_out/${T}_bench_rccl1_$i.json
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_rccl1_$i.json').read()); print('rccl world 1, a communicator per add in flight:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'single', (d.get('single_job') or {}).get('ms'))" | tee -a gpurun_out/${T}_rccl1.txt
done
tail -2 gpurun_out/${T}_bench_rccl1.err
echo "[$(( $(date +%s) - S0 )) s] rccl"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "two_rank" > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
echo "[$(( $(date +%s) - S0 )) s] tests"
sw() { local out; out=$(env $2 ZPQ_BENCH_NO_VARIANT=1 timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'steps', d['steps'], 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'), 'verified', d.get('verified_all_files'), {x:k[x] for x in list(k)[:4]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "16 lanes/chain (auto), 4 in flight" "X=1" "" | tee gpurun_out/${T}_extract.txt
sw "8 lanes/chain" "ZPQ_SHA256_GROUP=8" "" | tee -a gpurun_out/${T}_extract.txt
sw "4 lanes/chain" "ZPQ_SHA256_GROUP=4" "" | tee -a gpurun_out/${T}_extract.txt
sw "16 lanes/chain, 3 in flight" "X=1" "--pipeline 3" | tee -a gpurun_out/${T}_extract.txt
sw "8 lanes/chain, 3 in flight" "ZPQ_SHA256_GROUP=8" "--pipeline 3" | tee -a gpurun_out/${T}_extract.txt
sw "round-5 split, 4 in flight" "ZPQ_SHA256_GROUP=0" "" | tee -a gpurun_out/${T}_extract.txt
echo "[$(( $(date +%s) - S0 )) s] done"
tail -3 gpurun_out/${T}_last.err
