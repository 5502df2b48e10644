cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
