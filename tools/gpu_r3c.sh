# Round 3, second session: the twin-file fold on the GPU.  Priority order (the call may be cut short): parity tests of the
# new path, the headline with every check, extract, a few sweeps, kernel trace + FETCH/WRITE of the serial headline step.
# Every stage writes its own file under gpurun_out/ as soon as it ends.
R=$GRAFT_REPO_ROOT
T=${1:-r03c}
mkdir -p $R/gpurun_out
cd $R
export PYTHONUNBUFFERED=1
S0=$(date +%s)
el() { echo "[$(( $(date +%s) - S0 )) s] $*"; }
el start
timeout 400 python -m pytest tests/test_gpu_twins.py tests/test_gpu_verify.py tests/test_gpu_parity.py -k "twin or fragmenter or journaling or two_rank or verify or damaged or checksum" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_twins.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_tests_twins.log; tail -3 gpurun_out/${T}_tests_twins.log; el tests
timeout 300 python bench.py --workload silesia_x256_m1 > gpurun_out/${T}_bench_headline.json 2> gpurun_out/${T}_bench_headline.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${T}_bench_headline.json").read().strip().splitlines()[-1])
    print("headline", d["value"], d["ms_per_step"], "serial", d.get("ms_per_step_serial"), "depth", d.get("steps_in_flight"), {k:v for k,v in d.items() if k.startswith("verified")})
    print(" plain", d.get("every_byte_hashed")); print(" twins", {k:v for k,v in (d.get("twin_fold") or {}).items() if k!="note"})
    print(" roofline", d.get("roofline")); print(" kernels", d.get("kernels_ms_per_step")); print(" cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("headline failed", e); print(open("gpurun_out/${T}_bench_headline.err").read()[-1500:])
PY
el headline
timeout 300 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/${T}_bench_extract.json 2> gpurun_out/${T}_bench_extract.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${T}_bench_extract.json").read().strip().splitlines()[-1])
    print("extract", d["value"], d["ms_per_step"], "serial", d.get("ms_per_step_serial"), "depth", d.get("steps_in_flight"), {k:v for k,v in d.items() if k.startswith("verified")}, d.get("sha256_mismatches"))
    print(" twins", {k:v for k,v in (d.get("twin_fold") or {}).items() if k!="note"}); print(" kernels", d.get("kernels_ms_per_step")); print(" blake3", d.get("blake3_verify"))
except Exception as e:
    print("extract failed", e); print(open("gpurun_out/${T}_bench_extract.err").read()[-1500:])
PY
el extract
# sweeps of the headline (no verification, no CPU baseline, no second variant)
export ZPQ_BENCH_NO_PLAIN=1
B="python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 24"
sw() { # label, env, args
  local out; out=$(env $2 timeout 150 $B $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_serial'), d.get('steps_in_flight'))" 2>&1 | tail -1)
  echo "$1 | $2 | $3 | $out" | tee -a gpurun_out/${T}_sweep.txt; }
: > gpurun_out/${T}_sweep.txt
sw "depth 7" "X=1" "--pipeline 7"
sw "seg 2M depth 12" "ZPQ_LZ_SEG=2097152" "--pipeline 12"
# (a crossing-walk cap was swept here in the first run: slower, knob removed again)
sw "seg 2M depth 9" "ZPQ_LZ_SEG=2097152" "--pipeline 9"
el sweeps
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --pipeline 1"
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- $P --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/rocprof_stats.err
cd $R; python profiles/summarize.py gpurun_out $T > /dev/null 2>&1; cp profiles/${T}_rocprof_summary.txt gpurun_out/ 2>/dev/null; head -14 gpurun_out/${T}_rocprof_summary.txt; el trace
cd /tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_fetch.err
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- $P --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/rocprof_write.err
cd $R; python profiles/summarize.py gpurun_out $T > /dev/null 2>&1; cp profiles/${T}_rocprof_summary.txt profiles/traffic.json gpurun_out/ 2>/dev/null; el pmc
grep -A12 "FETCH_SIZE" gpurun_out/${T}_rocprof_summary.txt | head -16
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
rm -rf gpurun_out/prof_stats* gpurun_out/prof_fetch* gpurun_out/prof_write*
el done
