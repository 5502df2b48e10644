# Round 2, seventh GPU pass: decoder fast paths (parity + extract), deeper add pipelines with one hardware queue per stream.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 400 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "lz77_dec or lz77_decoder or resident or decompress or jidac or journaling or smoke or fixture" > gpurun_out/r2g_dec.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g_dec.log
timeout 300 python bench.py --workload extract_m1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_extract.json 2> gpurun_out/r2g_extract.err
for p in 5 6 7; do GPU_MAX_HW_QUEUES=24 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline $p > gpurun_out/r2g_q24_p$p.json 2> /dev/null; done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 6 > gpurun_out/r2g_q16_p6.json 2> /dev/null
tail -3 gpurun_out/r2g_dec.log
for f in r2g_extract r2g_q24_p5 r2g_q24_p6 r2g_q24_p7 r2g_q16_p6; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_serial"), {k:v for k,v in list(d["kernels_ms_per_step"].items())[:6]}, {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
