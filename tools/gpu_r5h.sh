# Round 5, final GPU call: the whole -m gpu suite, smoke(), and the default bench.py run on the final tree.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05k}
S0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider -x > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -4 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
echo "[$(( $(date +%s) - S0 )) s] tests"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "single", (d.get("single_job") or {}).get("ms"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("roofline", d.get("roofline"))
print("summary", d.get("workloads_summary"), d.get("failed_workloads"))
PY
bash tools/gpu_pmc_dup8.sh $T
echo "[$(( $(date +%s) - S0 )) s] done"
grep -v "^\[bench.py\]" gpurun_out/${T}_bench.err | tail -5
