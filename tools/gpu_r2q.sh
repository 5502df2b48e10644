# Round 2: LZ77 decoder token path (speculative parse + replay): A/B parity, extract bench
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_lzdec.py tests/test_gpu_sa.py -q -x -p no:cacheprovider > gpurun_out/r2q_dec.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2q_dec.log
tail -15 gpurun_out/r2q_dec.log
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "decode or decompress or extract or jidac or shim or fixture or unblock or resident" > gpurun_out/r2q_dec2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2q_dec2.log
tail -5 gpurun_out/r2q_dec2.log
timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline > gpurun_out/r2q_extract.json 2> gpurun_out/r2q_extract.err; echo "rc=$?"
ZPQ_LZDEC_SERIAL=1 timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline --steps 2 > gpurun_out/r2q_extract_serial.json 2> /dev/null; echo "rc=$?"
for f in r2q_extract r2q_extract_serial; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("$f", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
tail -3 gpurun_out/r2q_extract.err
