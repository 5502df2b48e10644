# Round 2: full GPU suite + extract with the new SHA-256 split
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2r_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2r_tests.log
tail -5 gpurun_out/r2r_tests.log
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/r2r_extract.json 2> gpurun_out/r2r_extract.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2r_extract.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")})
PY
