# Round 4: three waves per block on one table (lz77_waves.inc, default on): producer | evaluator | chain --
# parity on the chip and the bench lines
R=$GRAFT_REPO_ROOT
T=${1:-r04l}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -k "lz77 or compress_block or many_blocks or jidac or journaling or shim" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_lz.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_lz.log; tail -6 gpurun_out/${T}_tests_lz.log; echo "[$(( $(date +%s) - S0 )) s] tests"
: > gpurun_out/${T}_sweep.txt
sw() { # label, env, args
  local out; out=$(env $2 timeout 400 python bench.py --no-cpu-baseline $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:7]})" 2>&1 | tail -1)
  echo "[$(( $(date +%s) - S0 )) s] $1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; }
H="--workload silesia_x256_m1"
sw "dup8 duo"                  "X=1"            "--workload dup8_m1"
sw "headline serial duo"       "X=1"            "$H --steps 4 --pipeline 1"
sw "headline d6 duo"           "X=1"            "$H --steps 24"
sw "headline d6 duo seg 2M d12" "ZPQ_LZ_SEG=2097152" "$H --steps 48 --pipeline 12 --no-verify"
echo "[$(( $(date +%s) - S0 )) s] done"
