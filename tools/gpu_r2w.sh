# Round 2: verify path, two-rank text_m2, decode with 4 KiB segments
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_sa.py tests/test_gpu_lzdec.py -q -x -p no:cacheprovider > gpurun_out/r2w_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2w_tests.log
tail -25 gpurun_out/r2w_tests.log
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/r2w_extract.json 2> gpurun_out/r2w_extract.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2w_extract.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")})
PY
