# Round-end check: the whole GPU suite and the text_m2 line with its CPU baseline (the rest of the profile set: tools/gpu_profile.sh)
R=$GRAFT_REPO_ROOT
T=${1:-r02c}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
tail -5 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
timeout 600 python bench.py --workload text_m2 > gpurun_out/${T}_bench_text_m2.json 2> gpurun_out/${T}_bench_text_m2.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${T}_bench_text_m2.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified") or k.startswith("roundtrip")}, d.get("cpu_baseline",{}).get("value"))
PY
