# Round 4: profile, parity and dup8 after a change to the three-wave parse
R=$GRAFT_REPO_ROOT
T=${1:-r04r}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
timeout 300 python tools/lzprof2.py > gpurun_out/${T}_lzprof.txt 2>&1; cat gpurun_out/${T}_lzprof.txt; echo "[$(( $(date +%s) - S0 )) s] lzprof"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -k "lz77 or compress_block or many_blocks or jidac or journaling or shim" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_lz.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_lz.log; tail -4 gpurun_out/${T}_tests_lz.log; echo "[$(( $(date +%s) - S0 )) s] tests"
timeout 400 python bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('dup8', d['value'], d['ms_per_step'], {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:5]})" | tee gpurun_out/${T}_dup8.txt
echo "[$(( $(date +%s) - S0 )) s] done"
