# Quick GPU check used during development: the decoder and suffix-array suites plus two bench lines.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_lzdec.py tests/test_gpu_sa.py tests/test_gpu_verify.py -q -x -p no:cacheprovider > gpurun_out/small_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/small_tests.log
tail -6 gpurun_out/small_tests.log
timeout 600 python bench.py --workload text_m2 --no-cpu-baseline > gpurun_out/small_text_m2.json 2> gpurun_out/small_text_m2.err; echo "rc=$?"
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/small_extract.json 2> gpurun_out/small_extract.err; echo "rc=$?"
for f in small_text_m2 small_extract; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("$f", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified") or k.startswith("roundtrip")})
except Exception as e: print("ERR", e)
PY
done
