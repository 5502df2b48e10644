# Round profile set (run on the GPU box through gpurun): full -m gpu suite, rocprofv3 kernel trace + the two
# PMC passes of the serial bench, and the default bench line.  Outputs land in gpurun_out/.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/rocprof_stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --pipeline 1 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --pipeline 1 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/rocprof_write.err
cd $R
timeout 300 python bench.py --pipeline 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_serial.json 2> gpurun_out/bench_serial.err
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
