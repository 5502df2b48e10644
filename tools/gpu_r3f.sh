# Round 3: the chip sliced between the jobs in flight (ZPQ_CU_SLICES): do the long few-wave kernels stop stretching each other?
R=$GRAFT_REPO_ROOT
T=${1:-r03f}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
export ZPQ_BENCH_NO_PLAIN=1
B="python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 24"
sw() { # label, env, args
  local out; out=$(env $2 timeout 150 $B $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), 'chain', k.get('sha1_chain_kernel'), 'spec', k.get('lz77_spec_kernel'), 'resume', k.get('fragment_resume_kernel'), 'fspec', k.get('fragment_spec_kernel'), 'twin', k.get('twin_compare_kernel'))" 2>&1 | tail -1)
  echo "$1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; grep -m3 "zpaqhip" gpurun_out/${T}_last.err; }
: > gpurun_out/${T}_sweep.txt
while IFS='|' read -r label envs args; do
  [ -z "$label" ] && continue
  sw "$label" "$envs" "$args"
  el "$label"
done < tools/sweep_r3f.txt
el done
