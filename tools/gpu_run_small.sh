cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --durations=4 -k "cm_encode or cm_decode or cm_methods or generic or level5 or pcomp or decompresser" > gpurun_out/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4.log
