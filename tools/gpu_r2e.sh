# Round 2, fifth GPU pass: pipelined LZ77 parse (parity first), bench.py host-side variants.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -p no:cacheprovider -k "lz77 or compress_block or smoke or level1 or journaling or jidac or two_rank" > gpurun_out/r2e_lz.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e_lz.log
timeout 200 python bench.py --no-cpu-baseline --pipeline 1 --steps 3 --warmup 1 > gpurun_out/r2e_p1.json 2> gpurun_out/r2e_p1.err
timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2e_scatter.json 2> /dev/null
ZPQ_BENCH_TRAILER=loop timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2e_loop.json 2> /dev/null
ZPQ_BENCH_TRAILER=loop ZPQ_BENCH_CAT=1 timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2e_loop_cat.json 2> /dev/null
ZPQ_BENCH_TRAILER=loop ZPQ_BENCH_HOLD=1 timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2e_loop_hold.json 2> /dev/null
ZPQ_BENCH_HOLD=1 ZPQ_BENCH_CAT=1 timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2e_scatter_hold_cat.json 2> /dev/null
timeout 300 python bench.py --workload dup8_m1 --units 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2e_dup8_256.json 2> gpurun_out/r2e_dup8_256.err
tail -4 gpurun_out/r2e_lz.log
for f in r2e_p1 r2e_scatter r2e_loop r2e_loop_cat r2e_loop_hold r2e_scatter_hold_cat r2e_dup8_256; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_serial"), {k:v for k,v in list(d["kernels_ms_per_step"].items())[:6]}, {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
