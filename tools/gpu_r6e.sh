# Round 6, fifth GPU call: the communicator on a stream of its own (one rank through the multi-rank product path), the batched index
# read of zpqj_extract_dev, extract_m1 with twelve timed jobs (twice: run-to-run spread).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06e}
S0=$(date +%s)
timeout 900 python -m pytest tests/test_sharded_add.py tests/test_gpu_verify.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider -k "sharded or rccl or resident or verify or journaling or extract or jidac or two_rank or two_processes" > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -4 gpurun_out/${T}_tests.log
echo "[$(( $(date +%s) - S0 )) s] tests"
for i in 1 2; do
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 2> gpurun_out/${T}_bench_rccl1.err | tail -1 > gpurun_out/${T}_bench_rccl1_$i.json
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_rccl1_$i.json').read()); print('rccl world 1 through zpqj_add_sharded_dev:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'single', (d.get('single_job') or {}).get('ms'))" | tee -a gpurun_out/${T}_rccl1.txt
done
tail -2 gpurun_out/${T}_bench_rccl1.err
echo "[$(( $(date +%s) - S0 )) s] rccl"
for i in 1 2; do
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline 2> gpurun_out/${T}_extract.err | tail -1 > gpurun_out/${T}_bench_extract_$i.json
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_extract_$i.json').read()); print('extract_m1:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'steps', d['steps'], 'single', (d.get('single_job') or {}).get('ms'), 'fold on', (d.get('twin_fold_on') or {}).get('ms_per_step'), 'verified', d.get('verified_all_files'), 'roofline', (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic_over_algorithmic'))" | tee -a gpurun_out/${T}_extract.txt
done
echo "[$(( $(date +%s) - S0 )) s] done"
