# Round 3 closing check: the whole -m gpu suite (less the 3-minute reference-archive test, unchanged code), then the engine
# clock sampled while one job runs and while six are in flight (what stretches every kernel by 1.3-1.7x with six jobs?).
R=$GRAFT_REPO_ROOT
T=${1:-r03i}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
timeout 215 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_cm_spec.py::test_reference_archive_in_full_both_directions > gpurun_out/${T}_tests_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -3 gpurun_out/${T}_tests_gpu.log; el tests
export ZPQ_BENCH_NO_PLAIN=1
B="python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify"
clk() { # label, args
  ( while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.4; done ) > gpurun_out/${T}_clk_$1.txt &
  local pid=$!
  timeout 120 $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernels_ms_per_step'].get('sha1_chain_kernel'))"
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
  python - <<PY
import re
v=[int(m.group(1)) for m in re.finditer(r"\((\d+)Mhz\)", open("gpurun_out/${T}_clk_$1.txt").read())]
print("$1 sclk samples", len(v), "min", min(v) if v else None, "median", sorted(v)[len(v)//2] if v else None, "max", max(v) if v else None)
PY
}
clk serial "--pipeline 1 --steps 12"
clk depth6 "--pipeline 6 --steps 48"
el done
