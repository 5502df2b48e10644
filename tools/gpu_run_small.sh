cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "sha or lz77 or equals_oracle or fixture_blocks" > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t2.log
timeout 200 python bench.py --pipeline 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_serial.json 2> gpurun_out/b_stats.err
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/b_p3.err
