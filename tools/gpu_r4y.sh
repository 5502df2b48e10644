# Round 4, last GPU call: the code emission moved to the producer wave -- encoder parity on the chip, then config 4
R=$GRAFT_REPO_ROOT
cd $R
export PYTHONUNBUFFERED=1
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -k "lz77 or compress_block or many_blocks" -x -q -p no:cacheprovider > gpurun_out/r04y_tests_lz.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04y_tests_lz.log; tail -3 gpurun_out/r04y_tests_lz.log
timeout 130 python bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/r04y_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('dup8', d['value'], d['ms_per_step'], {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:3]})" | tee gpurun_out/r04y_dup8.txt
