# Round 4: segment size / depth of the headline with the three-wave walk (no verification: the pick is re-run verified)
R=$GRAFT_REPO_ROOT
T=${1:-r04p}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
: > gpurun_out/${T}_sweep.txt
sw() { # label, env, args
  local out; out=$(env $2 timeout 300 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), {a:k[a] for a in list(k)[:6]})" 2>&1 | tail -1)
  echo "[$(( $(date +%s) - S0 )) s] $1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; }
sw "seg 2M d8"   "ZPQ_LZ_SEG=2097152" "--steps 32 --pipeline 8"
sw "seg 2M d10"  "ZPQ_LZ_SEG=2097152" "--steps 40 --pipeline 10"
sw "seg 2M d14"  "ZPQ_LZ_SEG=2097152" "--steps 56 --pipeline 14"
sw "seg 4M d12"  "ZPQ_LZ_SEG=4194304" "--steps 48 --pipeline 12"
sw "seg 4M d16"  "ZPQ_LZ_SEG=4194304" "--steps 64 --pipeline 16"
sw "seg 4M d20"  "ZPQ_LZ_SEG=4194304" "--steps 80 --pipeline 20"
sw "seg 8M d16"  "ZPQ_LZ_SEG=8388608" "--steps 64 --pipeline 16"
sw "seg 1M d8"   "X=1" "--steps 32 --pipeline 8"
sw "seg 2M d12 q32" "ZPQ_LZ_SEG=2097152 GPU_MAX_HW_QUEUES=32" "--steps 48 --pipeline 12"
echo "[$(( $(date +%s) - S0 )) s] done"
