# Round 6, second GPU call: the SHA-256 group kernel and zpqj_extract_dev on the chip (tests), then extract_m1 through the product call
# with every restored byte hashed: lanes per chain 16 / 8 / 4 / 32 against the wave-wide + lane-wise split of round 5.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06b}
S0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_verify.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "sha256 or sha_ or resident_in_hbm or verify or extract" > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
echo "[$(( $(date +%s) - S0 )) s] tests"
sw() { local out; out=$(env $2 ZPQ_BENCH_NO_VARIANT=1 timeout 400 python bench.py --workload extract_m1 --no-cpu-baseline $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; a=d.get('kernels_ms_per_job_alone') or {}
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'), 'mism', d.get('sha256_mismatches'), 'verified', d.get('verified_all_files'), {x:k[x] for x in list(k)[:6]}, 'alone', {x:a[x] for x in list(a)[:5]})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "fold off, auto (16 lanes/chain)" "X=1" "" | tee gpurun_out/${T}_extract.txt
sw "fold off, 8 lanes/chain" "ZPQ_SHA256_GROUP=8" "" | tee -a gpurun_out/${T}_extract.txt
sw "fold off, 4 lanes/chain" "ZPQ_SHA256_GROUP=4" "" | tee -a gpurun_out/${T}_extract.txt
sw "fold off, 32 lanes/chain" "ZPQ_SHA256_GROUP=32" "" | tee -a gpurun_out/${T}_extract.txt
sw "fold off, round-5 split (wave-wide + lane-wise)" "ZPQ_SHA256_GROUP=0" "" | tee -a gpurun_out/${T}_extract.txt
sw "fold ON, auto" "X=1" "--twins" | tee -a gpurun_out/${T}_extract.txt
sw "fold off, auto, call-by-call (python)" "X=1" "--python-pipeline" | tee -a gpurun_out/${T}_extract.txt
echo "[$(( $(date +%s) - S0 )) s] done"
tail -5 gpurun_out/${T}_last.err
