# Round 3: cooperative wave placement (ZPQ_PLACE): parity with it forced on, the SIMD-key probe, the pipelined headline with and without.
R=$GRAFT_REPO_ROOT
T=${1:-r03g}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
: # (parity with ZPQ_PLACE=2: r03g first run, 43 passed)
:
export ZPQ_BENCH_NO_PLAIN=1
B="python bench.py --workload silesia_x256_m1 --no-cpu-baseline --steps 24"
sw() { # label, env, args
  local out; out=$(env $2 timeout 150 $B $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), 'chain', k.get('sha1_chain_kernel'), 'spec', k.get('lz77_spec_kernel'), 'resume', k.get('fragment_resume_kernel'), 'fspec', k.get('fragment_spec_kernel'), 'twin', k.get('twin_compare_kernel'), {a:b for a,b in d.items() if a.startswith('verified')})" 2>&1 | tail -1)
  echo "$1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; grep -m3 "zpaqhip" gpurun_out/${T}_last.err; }
: > gpurun_out/${T}_sweep.txt
while IFS='|' read -r label envs args; do
  [ -z "$label" ] && continue
  sw "$label" "$envs" "$args"
  el "$label"
done < tools/sweep_r3h_chain_cus_place.txt
el done
