# Round 2, sixth GPU pass: hardware-queue count vs steps in flight.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for q in 16 4 8; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2f_q$q.json 2> /dev/null; done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 4 > gpurun_out/r2f_q16_p4.json 2> /dev/null
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 5 > gpurun_out/r2f_q16_p5.json 2> /dev/null
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 2 > gpurun_out/r2f_q16_p2.json 2> /dev/null
for f in r2f_q16 r2f_q4 r2f_q8 r2f_q16_p4 r2f_q16_p5 r2f_q16_p2; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_serial"), {k:v for k,v in list(d["kernels_ms_per_step"].items())[:6]})
except Exception as e: print("ERR", e)
PY
done
