cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 --deselect tests/test_gpu_parity.py::test_sha_more_extents_than_lanes_longest_first > gpurun_out/t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t3.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/b3.json 2> gpurun_out/b3.err
