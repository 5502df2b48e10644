# Round 4, third GPU call: cheap headline explorations (no verification; every candidate default is re-run verified later) and
# the serial kernel list + FETCH / WRITE of the candidate-table path
R=$GRAFT_REPO_ROOT
T=${1:-r04d}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
: > gpurun_out/${T}_sweep.txt
sw() { # label, env, args
  local out; out=$(env $2 timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), {a:k[a] for a in list(k)[:7]})" 2>&1 | tail -1)
  echo "[$(( $(date +%s) - S0 )) s] $1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; }
CP="ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1"
sw "cand+pipe own sort d12"   "$CP ZPQ_SORT=own" "--steps 48 --pipeline 12"
sw "cand+pipe own sort d8"    "$CP ZPQ_SORT=own" "--steps 32 --pipeline 8"
sw "cand+pipe own sort d16"   "$CP ZPQ_SORT=own" "--steps 48 --pipeline 16"
sw "cand+pipe d12 seg 1M"     "$CP ZPQ_LZ_SEG=1048576" "--steps 48 --pipeline 12"
sw "cand+pipe own d12 seg 1M" "$CP ZPQ_SORT=own ZPQ_LZ_SEG=1048576" "--steps 48 --pipeline 12"
sw "default seg 2M d12"       "ZPQ_LZ_SEG=2097152" "--steps 48 --pipeline 12"
sw "default seg 4M d16"       "ZPQ_LZ_SEG=4194304" "--steps 48 --pipeline 16"
sw "default seg 8M d24"       "ZPQ_LZ_SEG=8388608" "--steps 72 --pipeline 24"
sw "default seg 4M d16 q32"   "ZPQ_LZ_SEG=4194304 GPU_MAX_HW_QUEUES=32" "--steps 48 --pipeline 16"
sw "cand+pipe own d12 q32"    "$CP ZPQ_SORT=own GPU_MAX_HW_QUEUES=32" "--steps 48 --pipeline 12"
sw "cand+pipe own d12 q8"     "$CP ZPQ_SORT=own GPU_MAX_HW_QUEUES=8" "--steps 48 --pipeline 12"
echo "[$(( $(date +%s) - S0 )) s] sweep done"
# serial kernel list + counters of the candidate path
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats* $R/gpurun_out/prof_fetch* $R/gpurun_out/prof_write*
P="python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --pipeline 1"
env $CP timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $P --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
env $CP timeout 150 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
env $CP timeout 150 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write.err
cd $R; cp profiles/traffic.json /tmp/traffic_keep.json; python profiles/summarize.py gpurun_out ${T}_cand > /dev/null 2>&1; cp profiles/${T}_cand_rocprof_summary.txt gpurun_out/; cp profiles/traffic.json gpurun_out/${T}_traffic_cand.json; cp /tmp/traffic_keep.json profiles/traffic.json
head -50 gpurun_out/${T}_cand_rocprof_summary.txt
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
echo "[$(( $(date +%s) - S0 )) s] done"
