# Round-end check: the whole GPU suite, smoke, and the bench lines most affected by the last changes (the full profile set: tools/gpu_profile.sh)
R=$GRAFT_REPO_ROOT
T=${1:-r02d}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log
tail -5 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
timeout 600 python bench.py --workload text_m2 > gpurun_out/${T}_bench_text_m2.json 2> gpurun_out/${T}_bench_text_m2.err; echo "rc=$?"
timeout 400 python bench.py --workload extract_m1 > gpurun_out/${T}_bench_extract.json 2> gpurun_out/${T}_bench_extract.err; echo "rc=$?"
timeout 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "rc=$?"
for f in ${T}_bench_text_m2 ${T}_bench_extract ${T}_bench; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("$f", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified") or k.startswith("roundtrip")}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("ERR", e)
PY
done
