# Round 6, the record runs on the final bench.py (eleven adds / five extracts in flight): the driver's command line and the default run
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06w}
S0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_cmd.json 2> gpurun_out/${T}_bench_driver_cmd.err; echo "driver-cmd bench rc=$?"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
grep "^\[bench.py\]" gpurun_out/${T}_bench.err > gpurun_out/${T}_bench_nested_lines.txt
for f in ${T}_bench_driver_cmd ${T}_bench; do python - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f: headline", d["value"], d["ms_per_step"], "cold", d.get("ms_per_step_cold"), "steps", d["steps"], "depth", d.get("steps_in_flight"), "single", (d.get("single_job") or {}).get("ms"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("  roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel"))
print("  summary", d.get("workloads_summary"), d.get("failed_workloads"))
e=d["workloads"]["extract_m1"]; print("  extract", e["value"], e["ms_per_step"], e.get("ms_per_step_cold"), e.get("steps_in_flight"), (e.get("twin_fold_on") or {}).get("ms_per_step"), (e.get("single_job") or {}).get("ms"))
PY
done
echo "[$(( $(date +%s) - S0 )) s] done"
grep -v "^\[bench.py\]" gpurun_out/${T}_bench.err | tail -3
