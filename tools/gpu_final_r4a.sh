# Round-4 closing run, part 1: the -m gpu suite and smoke()
R=$GRAFT_REPO_ROOT
T=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
tail -14 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
tail -2 gpurun_out/${T}_smoke.log
