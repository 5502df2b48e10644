# Round 5, last GPU call: the whole -m gpu suite on the final tree (no -x: every failure is listed).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05l}
timeout 1500 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -5 gpurun_out/${T}_tests_gpu.log
