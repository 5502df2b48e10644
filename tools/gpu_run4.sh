cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --durations=3 -k "e8e9" > gpurun_out/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4.log
