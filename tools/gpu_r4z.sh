# Round 4, last seconds of GPU: the waves' cycles per window with config 4's 1091 blocks in flight, then the one-workgroup-per-block parity test
cd $GRAFT_REPO_ROOT
T=${1:-r04z}
timeout 30 python tools/lzprof2.py 1 > gpurun_out/${T}_lzprof_full_chip.txt 2>&1; cat gpurun_out/${T}_lzprof_full_chip.txt
timeout 25 python -m pytest tests/test_gpu_round2.py -k segment_size_never_changes -x -q -p no:cacheprovider > gpurun_out/${T}_tests_lz.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_lz.log; tail -3 gpurun_out/${T}_tests_lz.log
