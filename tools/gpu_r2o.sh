# Round 2: suffix-array path v6 (packed LDS words, no bounds checks)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_sa.py -q -x -p no:cacheprovider > gpurun_out/r2o_sa.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o_sa.log
tail -5 gpurun_out/r2o_sa.log
for d in 1 2; do
timeout 600 python bench.py --workload text_m2 --steps 4 --warmup 1 --pipeline $d --no-cpu-baseline > gpurun_out/r2o_text_d$d.json 2> gpurun_out/r2o_text_d$d.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r2o_text_d$d.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_serial"], d["kernels_ms_per_step"])
print({k:v for k,v in d.items() if k.startswith("verified")})
PY
done
