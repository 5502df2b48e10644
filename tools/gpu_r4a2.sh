# Round 4, first GPU call: the experimental paths of round 3 on real hardware -- parity tests first, then A/B lines
# (trimmed from gpu_r4a.sh: most informative first, every line verifies its results).
R=$GRAFT_REPO_ROOT
T=${1:-r04a}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*" | tee -a gpurun_out/${T}_sweep.txt; }
: > gpurun_out/${T}_sweep.txt
ZPQ_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_lz_cand.py tests/test_gpu_cm_groups.py -q -p no:cacheprovider > gpurun_out/${T}_tests_experimental.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_experimental.log; tail -15 gpurun_out/${T}_tests_experimental.log; el tests
export ZPQ_BENCH_NO_PLAIN=1
sw() { # label, env, args
  local out; out=$(env $2 timeout 300 python bench.py --no-cpu-baseline $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'), {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:8]})" 2>&1 | tail -1)
  echo "[$(( $(date +%s) - S0 )) s] $1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; }
sw "headline default"            "X=1"                          "--workload silesia_x256_m1 --steps 24"
sw "headline cand+pipe depth 6"  "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1" "--workload silesia_x256_m1 --steps 24 --pipeline 6"
sw "headline cand+pipe depth 12" "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1" "--workload silesia_x256_m1 --steps 48 --pipeline 12"
sw "headline cand depth 6"       "ZPQ_LZ_CAND=1"                "--workload silesia_x256_m1 --steps 24 --pipeline 6"
sw "dup8 default"                "X=1"                          "--workload dup8_m1"
sw "dup8 cand+pipe"              "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1" "--workload dup8_m1"
sw "dup8 cand"                   "ZPQ_LZ_CAND=1"                "--workload dup8_m1"
sw "cm_m5 default"               "X=1"                          "--workload cm_m5"
sw "cm_m5 lane groups"           "ZPQ_CM_GROUPS=1"              "--workload cm_m5"
sw "cm_m5 lane groups, 16 waves" "ZPQ_CM_GROUPS=1 ZPQ_CM_WAVES=16" "--workload cm_m5"
sw "text_m2 default"             "X=1"                          "--workload text_m2"
sw "text_m2 own sort"            "ZPQ_SORT=own"                 "--workload text_m2"
sw "headline cand+pipe serial"   "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1" "--workload silesia_x256_m1 --steps 4 --pipeline 1"
sw "headline cand serial"        "ZPQ_LZ_CAND=1"                "--workload silesia_x256_m1 --steps 4 --pipeline 1"
sw "headline cand+pipe, own sort" "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1 ZPQ_SORT=own"   "--workload silesia_x256_m1 --steps 24 --pipeline 6"
sw "headline cand+pipe, shared sort arena, depth 16" "ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1 ZPQ_LZ_CAND_SHARED_SORT=1" "--workload silesia_x256_m1 --steps 48 --pipeline 16"
el done
