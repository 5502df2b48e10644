# Round 2, first GPU pass: new tests first (no -x: every failure is information), then the old suite, smoke, short benches.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 700 python -m pytest tests/test_gpu_round2.py -q --durations=8 -p no:cacheprovider > gpurun_out/r2a_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a_new.log
timeout 500 python -m pytest tests/test_gpu_parity.py -q --durations=6 -p no:cacheprovider > gpurun_out/r2a_old.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a_old.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2a_smoke.log 2>&1
timeout 300 python bench.py --steps 4 --warmup 1 > gpurun_out/r2a_bench_silesia.json 2> gpurun_out/r2a_bench_silesia.err
timeout 300 python bench.py --workload extract_m1 --steps 2 --warmup 1 > gpurun_out/r2a_bench_extract.json 2> gpurun_out/r2a_bench_extract.err
timeout 300 python bench.py --workload dup8_m1 --units 64 --steps 2 --warmup 1 > gpurun_out/r2a_bench_dup8_small.json 2> gpurun_out/r2a_bench_dup8_small.err
tail -3 gpurun_out/r2a_new.log; tail -3 gpurun_out/r2a_old.log; tail -1 gpurun_out/r2a_smoke.log
head -c 600 gpurun_out/r2a_bench_silesia.json; echo; tail -2 gpurun_out/r2a_bench_silesia.err
head -c 600 gpurun_out/r2a_bench_extract.json; echo; tail -2 gpurun_out/r2a_bench_extract.err
head -c 600 gpurun_out/r2a_bench_dup8_small.json; echo; tail -2 gpurun_out/r2a_bench_dup8_small.err
