cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "lz77 or equals_oracle or fixture_blocks or journaling or shim_multithreaded" > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t2.log
timeout 200 python bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_stats.json 2> gpurun_out/b_stats.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/b_p3.json 2> gpurun_out/b_p3.err
timeout 200 python tools/pcie_rate.py > gpurun_out/pcie.log 2>&1
