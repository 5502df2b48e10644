# Round 2: LZ77 batch budget = free - reserve (dup8 in one batch?), config 2 at six steps in flight
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --workload dup8_m1 --no-cpu-baseline > gpurun_out/r2x_dup8.json 2> gpurun_out/r2x_dup8.err; echo "rc=$?"
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r2x_bench.json 2> gpurun_out/r2x_bench.err; echo "rc=$?"
for f in r2x_dup8 r2x_bench; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("$f", d["value"], d["ms_per_step"], dict(list(d["kernels_ms_per_step"].items())[:6]), {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
tail -3 gpurun_out/r2x_dup8.err
