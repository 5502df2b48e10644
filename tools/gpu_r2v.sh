# Round 2: token replay v2 (gather/scatter rounds) + SHA-256 chain queue
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_lzdec.py tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "lzdec or token or decode or decompress or sha256 or extract or jidac or resident or stream or truncated or damaged or capacity or many" > gpurun_out/r2v_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2v_tests.log
tail -5 gpurun_out/r2v_tests.log
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/r2v_extract.json 2> gpurun_out/r2v_extract.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2v_extract.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")})
PY
