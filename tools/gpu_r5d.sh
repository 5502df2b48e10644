# Round 5, fourth GPU call: the tree as it stands (fragmenter at six waves per CU, product tail) -- kernel trace + HBM traffic of the
# headline job, the default bench.py run exactly as the driver starts it, the whole -m gpu suite, smoke().
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05d}
S0=$(date +%s)
bash tools/gpu_traffic.sh $T headline
echo "[$(( $(date +%s) - S0 )) s] traffic"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "single", d.get("single_job"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("roofline", d.get("roofline"))
print("summary", d.get("workloads_summary"))
for w,x in d.get("workloads",{}).items(): print(w, x.get("value"), x.get("ms_per_step"), x.get("single_job"), (x.get("roofline") or {}).get("traffic"), x.get("product_one_call"), x.get("error"))
PY
echo "[$(( $(date +%s) - S0 )) s] bench"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider -x > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -4 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
echo "[$(( $(date +%s) - S0 )) s] done"
tail -5 gpurun_out/${T}_bench.err
