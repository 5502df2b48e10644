# Round 4: what the chip does with twelve jobs in flight after the three-wave parse (kernel trace of the pipelined headline)
R=$GRAFT_REPO_ROOT
T=${1:-r04q}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
rm -rf $R/gpurun_out/prof_tl
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o r1 -- python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --no-kernel-timing --steps 48 --warmup 0 > $R/gpurun_out/${T}_tl.json 2> $R/gpurun_out/${T}_tl.err
python $R/profiles/timeline.py $R/gpurun_out/prof_tl 2500 300 > $R/gpurun_out/${T}_timeline_depth12.txt 2>> $R/gpurun_out/${T}_tl.err
head -42 $R/gpurun_out/${T}_timeline_depth12.txt; tail -c 400 $R/gpurun_out/${T}_tl.json | head -c 400
rm -rf $R/gpurun_out/prof_tl
