# Round 6, fourth GPU call: the whole -m gpu suite on the tree (no -x), smoke(), the driver's command line and the default bench.py run,
# one rank through the multi-rank product path again (collective order with lag depth/2), kernel traces + PMC traffic of the
# headline and of extract_m1 (new kernels), SQ counters of the context-mixing coder (the r03 file was three rounds old).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06d}
S0=$(date +%s)
timeout 1800 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -14 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
echo "[$(( $(date +%s) - S0 )) s] tests"
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 > gpurun_out/${T}_bench_rccl1.json 2> gpurun_out/${T}_bench_rccl1.err; echo "rccl1 rc=$?"
tail -1 gpurun_out/${T}_bench_rccl1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rccl world 1 through zpqj_add_sharded_dev:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'))"
echo "[$(( $(date +%s) - S0 )) s] rccl"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_cmd.json 2> gpurun_out/${T}_bench_driver_cmd.err; echo "driver-cmd bench rc=$?"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
grep "^\[bench.py\]" gpurun_out/${T}_bench.err > gpurun_out/${T}_bench_nested_lines.txt
for f in ${T}_bench_driver_cmd ${T}_bench; do python - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f: headline", d["value"], d["ms_per_step"], "cold", d.get("ms_per_step_cold"), "steps", d["steps"], "single", (d.get("single_job") or {}).get("ms"), "fold", d.get("value_twin_fold"), {k:v for k,v in d.items() if k.startswith("verified")})
print("  roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel"))
print("  summary", d.get("workloads_summary"), d.get("failed_workloads"))
PY
done
echo "[$(( $(date +%s) - S0 )) s] bench"
bash tools/gpu_traffic.sh $T headline extract_m1
bash tools/gpu_pmc_sq.sh $T cm_m5 --cm-blocks 2048 --cm-block-bytes 65536
bash tools/gpu_traffic.sh $T cm_m5:notrace
echo "[$(( $(date +%s) - S0 )) s] done"
grep -v "^\[bench.py\]" gpurun_out/${T}_bench.err | tail -5
