# Kernel trace of the method-2 path (four 64 MiB blocks) -> profiles/<tag>_rocprof_summary_text_m2.txt
R=$GRAFT_REPO_ROOT
T=${1:-r02e}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats_text_m2
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_text_m2 -o r1 -- python $R/bench.py --workload text_m2 --text-bytes 268435456 --pipeline 1 --no-cpu-baseline --no-verify --steps 2 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats_m2.err
cd $R
python profiles/summarize.py gpurun_out $T text_m2 | head -16
cp profiles/${T}_rocprof_summary_text_m2.txt gpurun_out/
rm -rf gpurun_out/prof_stats_text_m2
