# Round 2, second GPU pass: decoder / SHA-256 rework, sweeps, the full dup8 workload.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_round2.py -q -x --durations=5 -p no:cacheprovider -k "lz77 or sha256 or resident or e8e9 or compress_block or shim or jidac" > gpurun_out/r2b_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_new.log
timeout 200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "unsupported or sha" > gpurun_out/r2b_old.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_old.log
timeout 300 python bench.py --workload extract_m1 --steps 3 --warmup 1 > gpurun_out/r2b_bench_extract.json 2> gpurun_out/r2b_bench_extract.err
# SHA-256: all files lane-wise / all wave-wise / default split, to see what a chain costs in each shape
ZPQ_SHA256_CHAINS=0 timeout 300 python bench.py --workload extract_m1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/r2b_extract_lanes.json 2> /dev/null
ZPQ_SHA256_CHAINS=4096 timeout 300 python bench.py --workload extract_m1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/r2b_extract_chains.json 2> /dev/null
ZPQ_SHA256_CHAINS=512 timeout 300 python bench.py --workload extract_m1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/r2b_extract_c512.json 2> /dev/null
# add path: default (12 steps), deeper pipelines, SHA-1 waves per SIMD
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2b_silesia_p3.json 2> gpurun_out/r2b_silesia_p3.err
timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 4 > gpurun_out/r2b_silesia_p4.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 1 --steps 4 > gpurun_out/r2b_silesia_p1.json 2> /dev/null
for w in 1 3 4; do ZPQ_SHA_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 1 --steps 3 --warmup 1 > gpurun_out/r2b_silesia_p1_shaw$w.json 2> /dev/null; done
timeout 900 python bench.py --workload dup8_m1 --steps 2 --warmup 1 > gpurun_out/r2b_bench_dup8.json 2> gpurun_out/r2b_bench_dup8.err
tail -3 gpurun_out/r2b_new.log; tail -2 gpurun_out/r2b_old.log
for f in r2b_bench_extract r2b_extract_lanes r2b_extract_chains r2b_extract_c512 r2b_silesia_p3 r2b_silesia_p4 r2b_silesia_p1 r2b_silesia_p1_shaw1 r2b_silesia_p1_shaw3 r2b_silesia_p1_shaw4 r2b_bench_dup8; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:v for k,v in list(d["kernels_ms_per_step"].items())[:7]}, {k:v for k,v in d.items() if k.startswith("verified")})
except Exception as e: print("ERR", e)
PY
done
tail -3 gpurun_out/r2b_bench_dup8.err
