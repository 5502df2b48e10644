# Second GPU call of round 4 (prepared at the end of round 3): kernel trace and FETCH / WRITE counters of ONE serial headline
# step in the mode gpu_r4a.sh found best.  usage: tools/gpu_r4b.sh <tag> "<ENV=.. ENV=..>"   e.g.  r04b "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1"
R=$GRAFT_REPO_ROOT
T=${1:-r04b}
E=${2:-X=1}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --pipeline 1"
env $E timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $P --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
cd $R; python profiles/summarize.py gpurun_out $T > /dev/null 2>&1; cp profiles/${T}_rocprof_summary.txt gpurun_out/ 2>/dev/null; head -24 gpurun_out/${T}_rocprof_summary.txt; el trace
cd /tmp
env $E timeout 150 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
env $E timeout 150 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write.err
cd $R; python profiles/summarize.py gpurun_out $T > /dev/null 2>&1; cp profiles/${T}_rocprof_summary.txt profiles/traffic.json gpurun_out/ 2>/dev/null; el pmc
grep -A14 "FETCH_SIZE" gpurun_out/${T}_rocprof_summary.txt | head -18
# the whole default bench line in that mode (what the driver would see if it became the default)
env $E timeout 400 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; tail -c 600 gpurun_out/${T}_bench_default.json; el bench
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
el done
