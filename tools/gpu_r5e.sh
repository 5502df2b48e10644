# Round 5, fifth GPU call: the default bench.py run exactly as the driver starts it (after the fix of config 4's product call).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05e}
S0=$(date +%s)
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "single", (d.get("single_job") or {}).get("ms"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("summary", d.get("workloads_summary"), d.get("failed_workloads"))
for w,x in d.get("workloads",{}).items(): print(w, x.get("value"), x.get("ms_per_step"), (x.get("single_job") or {}).get("ms"), (x.get("roofline") or {}).get("traffic"), x.get("product_one_call"), {k:v for k,v in x.items() if k.startswith("verified")}, x.get("error"))
PY
echo "[$(( $(date +%s) - S0 )) s] bench"
grep -v "^\[bench.py\]" gpurun_out/${T}_bench.err | tail -5
