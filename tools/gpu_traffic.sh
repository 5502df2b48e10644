# HBM traffic and kernel traces of the bench workloads (run on the GPU box through gpurun):  bash tools/gpu_traffic.sh <tag> <workload>...
# Per workload three runs of the same command: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes:
# the two counters do not fit one, MI355X_MICROARCH.md); profiles/summarize.py turns them into profiles/<tag>_rocprof_summary[_<workload>].txt
# and profiles/traffic[_<workload>].json (bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE), which bench.py's roofline.traffic reads.
# <workload>:notrace skips the kernel trace (an older summary of that workload is kept).  "headline" = silesia_x256_m1 as one job at a time (--pipeline 1), every byte hashed (the default).
R=$GRAFT_REPO_ROOT
T=$1; shift
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1
for W in "$@"; do
  S0=$(date +%s)
  W0=$W; W=${W%:notrace}
  case $W in
    headline) SFX=""; ARGS="--workload silesia_x256_m1 --pipeline 1";;
    cm_m5) SFX="_$W"; ARGS="--workload $W --cm-blocks 2048 --cm-block-bytes 65536";;      # (bench.py scales traffic_cm_m5.json from this size)
    *) SFX="_$W"; ARGS="--workload $W --pipeline 1";;
  esac
  rm -rf $R/gpurun_out/prof_stats$SFX $R/gpurun_out/prof_fetch$SFX $R/gpurun_out/prof_write$SFX
  [ "$W0" = "$W" ] && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats$SFX -o r1 -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/${T}_bench_under_rocprof$SFX.json 2> $R/gpurun_out/${T}_rocprof_stats$SFX.err
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch$SFX -o r1 -- python $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/${T}_rocprof_fetch$SFX.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write$SFX -o r1 -- python $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/${T}_rocprof_write$SFX.err
  (cd $R && python profiles/summarize.py gpurun_out $T ${SFX#_} > /dev/null 2> gpurun_out/${T}_summarize$SFX.err)
  cp $R/profiles/${T}_rocprof_summary$SFX.txt $R/profiles/traffic$SFX.json $R/gpurun_out/ 2>/dev/null
  rm -rf $R/gpurun_out/prof_stats$SFX $R/gpurun_out/prof_fetch$SFX $R/gpurun_out/prof_write$SFX
  echo "[$(( $(date +%s) - S0 )) s] $W"; head -8 $R/gpurun_out/${T}_rocprof_summary$SFX.txt
done
