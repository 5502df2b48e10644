# Round 5, sixth GPU call: the fragment SHA-1 pass -- staging rows parked late (ZPQ_SHA1_LATE=1) and waves per CU -- in the pipelined headline.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1
T=${1:-r05f}
S0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_round2.py -m gpu -k "sha1_extents" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_sha1.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_sha1.log; tail -3 gpurun_out/${T}_tests_sha1.log
sw() { local out; out=$(env $2 timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 36 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; a=d.get('kernels_ms_per_job_alone') or {}
print(d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), {x:k[x] for x in k if 'frag' in x or 'sha1_ext' in x}, 'alone', {x:a[x] for x in a if 'frag' in x or 'sha1_ext' in x})" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
sw "default (rows parked after the first block)" "ZPQ_SHA1_LATE=0" | tee gpurun_out/${T}_sweep_sha1.txt
sw "rows parked after both blocks" "ZPQ_SHA1_LATE=1" | tee -a gpurun_out/${T}_sweep_sha1.txt
sw "late, 4 waves/CU" "ZPQ_SHA1_LATE=1 ZPQ_SHA_WAVES=1" | tee -a gpurun_out/${T}_sweep_sha1.txt
sw "early, 4 waves/CU" "ZPQ_SHA1_LATE=0 ZPQ_SHA_WAVES=1" | tee -a gpurun_out/${T}_sweep_sha1.txt
sw "default again" "ZPQ_SHA1_LATE=0" | tee -a gpurun_out/${T}_sweep_sha1.txt
echo "[$(( $(date +%s) - S0 )) s] done"
tail -3 gpurun_out/${T}_last.err
