# Round profile set (run on the GPU box through gpurun): full -m gpu suite, smoke, rocprofv3 kernel trace + the two
# HBM-traffic PMC passes of the serial add step, and the bench lines of every workload.  Outputs land in gpurun_out/.
R=$GRAFT_REPO_ROOT
T=${1:-r02}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/${T}_bench_under_rocprof.json 2> $R/gpurun_out/rocprof_stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --pipeline 1 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --pipeline 1 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/rocprof_write.err
# method 2 (256 MiB of text: four 64 MiB blocks) and extract: kernel trace; method 2 also the two traffic passes
rm -rf $R/gpurun_out/prof_stats_text_m2 $R/gpurun_out/prof_fetch_text_m2 $R/gpurun_out/prof_write_text_m2 $R/gpurun_out/prof_stats_extract_m1
M2="--workload text_m2 --text-bytes 268435456 --pipeline 1 --no-cpu-baseline --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_text_m2 -o r1 -- python $R/bench.py $M2 --steps 2 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats_m2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch_text_m2 -o r1 -- python $R/bench.py $M2 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch_m2.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write_text_m2 -o r1 -- python $R/bench.py $M2 --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write_m2.err
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_extract_m1 -o r1 -- python $R/bench.py --workload extract_m1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/rocprof_stats_ex.err
cd $R
python profiles/summarize.py gpurun_out $T > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T text_m2 > /dev/null 2>&1
python profiles/summarize.py gpurun_out $T extract_m1 > /dev/null 2>&1
cp profiles/${T}_rocprof_summary*.txt profiles/traffic*.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
timeout 300 python bench.py --pipeline 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_serial.json 2> gpurun_out/${T}_bench_serial.err
timeout 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python bench.py --force-collectives --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_rccl1.json 2> gpurun_out/${T}_bench_rccl1.err
timeout 400 python bench.py --workload extract_m1 > gpurun_out/${T}_bench_extract.json 2> gpurun_out/${T}_bench_extract.err
timeout 900 python bench.py --workload dup8_m1 > gpurun_out/${T}_bench_dup8.json 2> gpurun_out/${T}_bench_dup8.err
timeout 600 python bench.py --workload text_m2 > gpurun_out/${T}_bench_text_m2.json 2> gpurun_out/${T}_bench_text_m2.err
tail -4 gpurun_out/${T}_tests_gpu.log; tail -1 gpurun_out/${T}_smoke.log
for f in ${T}_bench_serial ${T}_bench ${T}_bench_rccl1 ${T}_bench_extract ${T}_bench_dup8 ${T}_bench_text_m2; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("ERR", e)
PY
done
tail -2 gpurun_out/${T}_bench_rccl1.err
