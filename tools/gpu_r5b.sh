# Round 5, second GPU call: the tree with the token-per-lane emitter and the literal bytes packed by the token kernel; the product's
# zpqj_add_dev as the timed step; HBM traffic of every workload; the fragmenter's crossing budget with every byte hashed.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05b}
S0=$(date +%s)
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('$1', d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), 'fold_on', (d.get('twin_fold_on') or {}).get('ms_per_step'), {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:7]}, 'roofline', d.get('roofline'))"; }
timeout 120 python -m pytest tests/test_gpu_round2.py -m gpu -k "add_dev" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_add_dev.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_add_dev.log; tail -3 gpurun_out/${T}_tests_add_dev.log
timeout 400 python bench.py --no-cpu-baseline --workload silesia_x256_m1 2>gpurun_out/${T}_err1.txt | tail -1 | tee gpurun_out/${T}_headline_product.json | line headline_product
timeout 400 python bench.py --no-cpu-baseline --workload silesia_x256_m1 --python-pipeline 2>gpurun_out/${T}_err2.txt | tail -1 | tee gpurun_out/${T}_headline_python.json | line headline_python
echo "[$(( $(date +%s) - S0 )) s] headline"
# the fragmenter's crossing budget and resume grid, every byte hashed, twelve jobs in flight (no verification: the GPU tests pin the result for any budget)
export ZPQ_BENCH_NO_VARIANT=1
sw() { local out; out=$(env $2 timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 36 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; a=d.get('kernels_ms_per_job_alone') or {}
print(d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in k if 'frag' in x}, 'alone', {x:a[x] for x in a if 'frag' in x or 'sha1_ext' in x})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "budget 256K (default)" "X=1" | tee gpurun_out/${T}_sweep_frag_budget.txt
sw "budget 128K" "ZPQ_FRAG_BUDGET=131072" | tee -a gpurun_out/${T}_sweep_frag_budget.txt
sw "budget 64K" "ZPQ_FRAG_BUDGET=65536" | tee -a gpurun_out/${T}_sweep_frag_budget.txt
sw "budget 64K, resume 4 waves/CU" "ZPQ_FRAG_BUDGET=65536 ZPQ_FRAG_RESUME_WAVES=4" | tee -a gpurun_out/${T}_sweep_frag_budget.txt
sw "budget 32K, resume 4 waves/CU" "ZPQ_FRAG_BUDGET=32768 ZPQ_FRAG_RESUME_WAVES=4" | tee -a gpurun_out/${T}_sweep_frag_budget.txt
echo "[$(( $(date +%s) - S0 )) s] sweep"
export -n ZPQ_BENCH_NO_VARIANT
timeout 300 python bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/${T}_err3.txt | tail -1 | tee gpurun_out/${T}_dup8.json | line dup8
echo "[$(( $(date +%s) - S0 )) s] dup8"
bash tools/gpu_traffic.sh $T headline dup8_m1 extract_m1:notrace text_m2:notrace
echo "[$(( $(date +%s) - S0 )) s] done"
for f in 1 2 3; do [ -s gpurun_out/${T}_err$f.txt ] && { echo "== err$f"; tail -5 gpurun_out/${T}_err$f.txt; }; done
tail -3 gpurun_out/${T}_last.err
