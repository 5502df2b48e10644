# Quick GPU check used during development: the suffix-array and decoder suites plus the text_m2 line.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_sa.py tests/test_gpu_lzdec.py -q -x -p no:cacheprovider > gpurun_out/small_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/small_tests.log
tail -4 gpurun_out/small_tests.log
timeout 600 python bench.py --workload text_m2 --no-cpu-baseline > gpurun_out/small_text_m2.json 2> gpurun_out/small_text_m2.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/small_text_m2.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_serial"], {k:v for k,v in list(d["kernels_ms_per_step"].items())[:6]}, {k:v for k,v in d.items() if k.startswith("verified") or k.startswith("roundtrip")})
PY
