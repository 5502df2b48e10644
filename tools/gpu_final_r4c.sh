# Round-4 closing run, part 3: rocprofv3 kernel trace + stats of THE command bench.py's headline line comes from (twelve jobs in
# flight, kernel timing on), summarised per kernel and as a timeline; then the -m gpu suite once more on the final tree
R=$GRAFT_REPO_ROOT
T=${1:-r04}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
rm -rf $R/gpurun_out/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --warmup 0 > $R/gpurun_out/${T}_headline_traced.json 2> $R/gpurun_out/rocprof_stats.err
cd $R
python profiles/timeline.py gpurun_out/prof_stats 3000 300 > gpurun_out/${T}_timeline_headline_depth12.txt 2>> gpurun_out/rocprof_stats.err
head -30 gpurun_out/${T}_timeline_headline_depth12.txt
python -c "
import json; d=json.load(open('gpurun_out/${T}_headline_traced.json')); print('traced run:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline_chip_filling']['kernel'], d['roofline_chip_filling']['avg_launch_ms'])"
rm -rf gpurun_out/prof_stats
export -n ZPQ_BENCH_NO_PLAIN
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
tail -4 gpurun_out/${T}_tests_gpu.log
