# Round 5: the context-mixing tests after the second pass through the interpreter kernel went in (host-side change in cm.hip / cm_jit.hip).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05m}
timeout 500 python -m pytest tests/test_gpu_cm_spec.py tests/test_gpu_segments.py tests/test_gpu_parity.py -m gpu -k "cm or level5 or gives_up or segment or fixture_arch or builtin or methods" -q -p no:cacheprovider > gpurun_out/${T}_tests_cm.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_cm.log; tail -4 gpurun_out/${T}_tests_cm.log
