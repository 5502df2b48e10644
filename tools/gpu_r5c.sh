# Round 5, third GPU call: where a product add spends its host time (ZPQJ_TIMING), the fragmenter's occupancy (waves per CU -> LDS held,
# segment size, crossing share) with every byte hashed, config 4 with its archive verified, the RCCL collective with one rank.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05c}
S0=$(date +%s)
timeout 200 python -m pytest tests/test_sharded_add.py -m gpu -k "rccl" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_rccl.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_rccl.log; tail -3 gpurun_out/${T}_tests_rccl.log
export ZPQ_BENCH_NO_VARIANT=1
for M in "" "--twins"; do
  ZPQJ_TIMING=1 timeout 200 python bench.py --workload silesia_x256_m1 --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline --no-verify $M 2>gpurun_out/${T}_timing.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('product alone $M', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
  grep "zpqj add" gpurun_out/${T}_timing.err | tail -2
  timeout 200 python bench.py --workload silesia_x256_m1 --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --python-pipeline $M 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('python alone $M', d['value'], d['ms_per_step'])"
done
echo "[$(( $(date +%s) - S0 )) s] timing"
sw() { local out; out=$(env $2 timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 36 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; a=d.get('kernels_ms_per_job_alone') or {}
print(d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in k if 'frag' in x or 'sha1_ext' in x or 'lz77_spec' in x}, 'alone', {x:a[x] for x in a if 'frag' in x or 'sha1_ext' in x})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "10 waves/CU (default)" "X=1" | tee gpurun_out/${T}_sweep_frag_waves.txt
sw "8 waves/CU" "ZPQ_FRAG_WAVES=8" | tee -a gpurun_out/${T}_sweep_frag_waves.txt
sw "6 waves/CU" "ZPQ_FRAG_WAVES=6" | tee -a gpurun_out/${T}_sweep_frag_waves.txt
sw "4 waves/CU" "ZPQ_FRAG_WAVES=4" | tee -a gpurun_out/${T}_sweep_frag_waves.txt
sw "6 waves/CU, budget 64K" "ZPQ_FRAG_WAVES=6 ZPQ_FRAG_BUDGET=65536" | tee -a gpurun_out/${T}_sweep_frag_waves.txt
sw "10 waves/CU, fold on" "X=1" "--twins" | tee -a gpurun_out/${T}_sweep_frag_waves.txt
echo "[$(( $(date +%s) - S0 )) s] sweep"
export -n ZPQ_BENCH_NO_VARIANT
timeout 400 python bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/${T}_err3.txt | tail -1 | tee gpurun_out/${T}_dup8.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dup8', d['value'], d['ms_per_step'], {a:b for a,b in d.items() if a.startswith('verified')}, d.get('archive_blocks'), d.get('roofline'))"
echo "[$(( $(date +%s) - S0 )) s] dup8"
bash tools/gpu_traffic.sh $T text_m2:notrace
echo "[$(( $(date +%s) - S0 )) s] done"
[ -s gpurun_out/${T}_err3.txt ] && tail -5 gpurun_out/${T}_err3.txt
tail -3 gpurun_out/${T}_last.err
