# Round 6: the record runs with eleven jobs in flight: the driver's command line, the default run, one rank through the multi-rank path
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06l}
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
PY
done
echo "[$(( $(date +%s) - S0 )) s] bench"
for i in 1 2; do
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 2> gpurun_out/${T}_bench_rccl1.err | tail -1 > gpurun_out/${T}_bench_rccl1.json
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_rccl1.json').read()); print('rccl world 1:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'))" | tee -a gpurun_out/${T}_rccl1.txt
done
echo "[$(( $(date +%s) - S0 )) s] done"
