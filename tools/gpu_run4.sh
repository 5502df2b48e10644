cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ZPQ_CM_STATS=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "level5_prefix" > gpurun_out/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4.log
