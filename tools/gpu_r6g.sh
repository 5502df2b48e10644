# Round 6, final GPU call: the whole -m gpu suite on the final tree (no -x), smoke(), the driver's command line and the default bench.py
# run, one rank through the multi-rank product path, kernel trace + PMC traffic of extract_m1 (eight lanes per chain).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06g}
S0=$(date +%s)
timeout 1800 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -12 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
echo "[$(( $(date +%s) - S0 )) s] tests"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_cmd.json 2> gpurun_out/${T}_bench_driver_cmd.err; echo "driver-cmd bench rc=$?"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
grep "^\[bench.py\]" gpurun_out/${T}_bench.err > gpurun_out/${T}_bench_nested_lines.txt
for f in ${T}_bench_driver_cmd ${T}_bench; do python - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f: headline", d["value"], d["ms_per_step"], "cold", d.get("ms_per_step_cold"), "steps", d["steps"], "single", (d.get("single_job") or {}).get("ms"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("  roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel"))
print("  summary", d.get("workloads_summary"), d.get("failed_workloads"))
e=d["workloads"]["extract_m1"]; print("  extract", e["value"], e["ms_per_step"], e.get("ms_per_step_cold"), (e.get("twin_fold_on") or {}).get("ms_per_step"), (e.get("single_job") or {}).get("ms"), e.get("roofline"))
PY
done
echo "[$(( $(date +%s) - S0 )) s] bench"
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 2> gpurun_out/${T}_bench_rccl1.err | tail -1 > gpurun_out/${T}_bench_rccl1.json
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_rccl1.json').read()); print('rccl world 1:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'single', (d.get('single_job') or {}).get('ms'))"
bash tools/gpu_traffic.sh $T extract_m1
echo "[$(( $(date +%s) - S0 )) s] done"
grep -v "^\[bench.py\]" gpurun_out/${T}_bench.err | tail -5
