# Round 2: direct LZ77 mode (parity, dup8), RCCL single-rank path with a watchdog.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "lz77 or compress_block or smoke or level1 or journaling or jidac or two_rank or shim" > gpurun_out/r2h_lz.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h_lz.log
ZPQ_BENCH_WATCHDOG=150 NCCL_DEBUG=WARN timeout 200 python bench.py --force-collectives --steps 4 --warmup 1 --no-cpu-baseline --pipeline 3 > gpurun_out/r2h_rccl1.json 2> gpurun_out/r2h_rccl1.err
timeout 900 python bench.py --workload dup8_m1 --no-cpu-baseline > gpurun_out/r2h_dup8.json 2> gpurun_out/r2h_dup8.err
ZPQ_LZ_DIRECT=0 timeout 300 python bench.py --workload dup8_m1 --units 256 --no-cpu-baseline --no-verify > gpurun_out/r2h_dup8_256_spec.json 2> /dev/null
ZPQ_LZ_DIRECT=1 timeout 300 python bench.py --workload dup8_m1 --units 256 --no-cpu-baseline --no-verify > gpurun_out/r2h_dup8_256_direct.json 2> /dev/null
tail -3 gpurun_out/r2h_lz.log
for f in r2h_rccl1 r2h_dup8 r2h_dup8_256_spec r2h_dup8_256_direct; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:v for k,v in list(d["kernels_ms_per_step"].items())[:6]}, {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
tail -40 gpurun_out/r2h_rccl1.err
