# Round 2: suffix-array path (first run) + sanity of the committed bench.py
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_sa.py -q -x -p no:cacheprovider > gpurun_out/r2i_sa.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i_sa.log
tail -30 gpurun_out/r2i_sa.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; tail -c 1500 gpurun_out/r2i_bench.json
