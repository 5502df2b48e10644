# Round 5, seventh GPU call: shallower LDS rings in the LZ77 parse workgroups (ring 1 / ring 2 of 6 / 5 windows = 31 KB per workgroup;
# variants built by tools/make_variant.sh with -DZPQ_TRIO_R1 / -DZPQ_TRIO_R2) -- headline in flight, config 4, encoder parity subset.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1
T=${1:-r05g}
S0=$(date +%s)
LZ='lz77 or compress_block or many_blocks'
sw() { local out; out=$($2 bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 36 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; a=d.get('kernels_ms_per_job_alone') or {}
print(d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in k if 'lz77' in x or 'frag' in x}, 'alone', {x:a[x] for x in a if 'lz77' in x})" 2>&1 | tail -1); echo "$1 | $out"; }
sw "rings 6 / 5 (tree)" "timeout 200 python" | tee gpurun_out/${T}_sweep_rings.txt
sw "rings 4 / 3" "timeout 200 python tools/run_variant.py ring43" | tee -a gpurun_out/${T}_sweep_rings.txt
sw "rings 3 / 2" "timeout 200 python tools/run_variant.py ring32" | tee -a gpurun_out/${T}_sweep_rings.txt
sw "rings 6 / 5 (tree) again" "timeout 200 python" | tee -a gpurun_out/${T}_sweep_rings.txt
for V in ring43 ring32; do
  timeout 150 python tools/run_variant.py $V -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -k "$LZ" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_$V.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_$V.log; tail -2 gpurun_out/${T}_tests_$V.log
  timeout 250 python tools/run_variant.py $V bench.py --no-cpu-baseline --no-verify --workload dup8_m1 2>>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dup8 $V', d['value'], d['ms_per_step'], d['kernels_ms_per_step'].get('lz77_direct_kernel'))" | tee -a gpurun_out/${T}_sweep_rings.txt
done
echo "[$(( $(date +%s) - S0 )) s] done"
tail -3 gpurun_out/${T}_last.err
