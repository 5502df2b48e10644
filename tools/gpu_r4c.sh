# Round 4, second GPU call: (1) the coder / post-processor tests after the JIT cache hardening, (2) what the chip does with
# six / twelve jobs in flight: rocprofv3 kernel traces of the pipelined headline, analysed by profiles/timeline.py
R=$GRAFT_REPO_ROOT
T=${1:-r04c}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
timeout 400 python -m pytest tests/test_gpu_cm_spec.py tests/test_gpu_cm.py -x -q -p no:cacheprovider > gpurun_out/${T}_tests_cm.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_cm.log; tail -5 gpurun_out/${T}_tests_cm.log; el tests
cd /tmp; export TMPDIR=/tmp
for mode in default cand12; do
  if [ $mode = default ]; then E="X=1"; A="--pipeline 6 --steps 18"; else E="ZPQ_LZ_CAND=1 ZPQ_LZ_CAND_PIPE=1"; A="--pipeline 12 --steps 36"; fi
  rm -rf $R/gpurun_out/prof_tl
  env $E timeout 240 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o r1 -- python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --no-kernel-timing $A --warmup 0 > $R/gpurun_out/${T}_tl_${mode}.json 2> $R/gpurun_out/${T}_tl_${mode}.err
  python $R/profiles/timeline.py $R/gpurun_out/prof_tl 1500 250 > $R/gpurun_out/${T}_timeline_${mode}.txt 2>> $R/gpurun_out/${T}_tl_${mode}.err
  head -40 $R/gpurun_out/${T}_timeline_${mode}.txt; tail -c 300 $R/gpurun_out/${T}_tl_${mode}.json; el $mode
done
rm -rf $R/gpurun_out/prof_tl
el done
