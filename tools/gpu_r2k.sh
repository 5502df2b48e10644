# Round 2: suffix-array path v2 (SA-order candidates, speculative chain) -- parity + text_m2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_sa.py -q -x -p no:cacheprovider > gpurun_out/r2k_sa.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k_sa.log
tail -15 gpurun_out/r2k_sa.log
timeout 600 python bench.py --workload text_m2 --steps 2 --warmup 1 > gpurun_out/r2k_text.json 2> gpurun_out/r2k_text.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2k_text.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"], d["kernels_ms_per_step"])
print({k:v for k,v in d.items() if k.startswith("verified")}, d.get("cpu_baseline"))
PY
tail -5 gpurun_out/r2k_text.err
