# Round 2: suffix-array path v3 (BWT-byte tile) + sanity of the other workloads with the committed bench.py
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_sa.py -q -x -p no:cacheprovider > gpurun_out/r2l_sa.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l_sa.log
tail -5 gpurun_out/r2l_sa.log
timeout 600 python bench.py --workload text_m2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2l_text.json 2> gpurun_out/r2l_text.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2l_text.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"])
print({k:v for k,v in d.items() if k.startswith("verified")})
PY
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/r2l_extract.json 2> gpurun_out/r2l_extract.err; echo "rc=$?"
timeout 300 python bench.py --workload dup8_m1 --units 64 --no-cpu-baseline > gpurun_out/r2l_dup8_64.json 2> gpurun_out/r2l_dup8_64.err; echo "rc=$?"
ZPQ_BENCH_WATCHDOG=200 timeout 300 python bench.py --force-collectives --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2l_rccl1.json 2> gpurun_out/r2l_rccl1.err; echo "rc=$?"
for f in r2l_extract r2l_dup8_64 r2l_rccl1; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["roofline"], {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
