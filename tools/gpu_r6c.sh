# Round 6, third GPU call: (1) tests of this round's host-path changes on the chip (sharded add with files resident in HBM over gloo and over
# the in-tree RCCL collectives, two-rank bench through the product call, HCOMP credit, pool trim, add_dev argument checks);
# (2) bench.py --force-collectives: one rank through the multi-rank code path (zpqj_add_sharded_dev + zpqr_allgatherv[_dev], RCCL world 1);
# (3) context mixing: what a block costs per byte at 0.25 / 0.5 / 1 / 2 waves per SIMD and where the cycles go (ZPQ_CM_PROF) -- the
#     numbers behind the LDS-residency decision.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06c}
S0=$(date +%s)
timeout 1500 python -m pytest tests/test_sharded_add.py tests/test_gpu_verify.py "tests/test_gpu_parity.py::test_two_rank_add_is_bit_identical_to_serial" "tests/test_gpu_parity.py::test_two_rank_shared_corpus_equals_the_single_gpu_archive" tests/test_gpu_cm_spec.py -m gpu -q -p no:cacheprovider --durations=6 -k "not reference_archive_in_full" > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -12 gpurun_out/${T}_tests.log
timeout 300 python -m pytest tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider -k "add_dev or sha256" >> gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
echo "[$(( $(date +%s) - S0 )) s] tests"
ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 > gpurun_out/${T}_bench_rccl1.json 2> gpurun_out/${T}_bench_rccl1.err; echo "rccl1 rc=$?"
tail -1 gpurun_out/${T}_bench_rccl1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rccl world 1 through zpqj_add_sharded_dev:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), d['timed_step'][:60], d['config'].get('blocks'), d['config'].get('out_bytes'))"; tail -3 gpurun_out/${T}_bench_rccl1.err
echo "[$(( $(date +%s) - S0 )) s] rccl"
for NB in 256 512 1024 2048; do
  ZPQ_CM_PROF=1 ZPQ_JIT_NOCACHE=1 timeout 300 python bench.py --workload cm_m5 --cm-blocks $NB --cm-block-bytes 65536 --no-cpu-baseline --no-verify --steps 1 --warmup 1 2> gpurun_out/${T}_cm_$NB.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cm_m5 blocks $NB:', 'in MB/s', d['input_MBps'], 'waves/SIMD', d['waves_per_simd'], 'KB/s per block', d['input_KBps_per_block'], 'ms', d['ms_per_step'])" | tee -a gpurun_out/${T}_cm_occupancy.txt
  grep "cm prof" gpurun_out/${T}_cm_$NB.err | tail -1 | tee -a gpurun_out/${T}_cm_occupancy.txt
done
for NB in 512 2048; do
  timeout 300 python bench.py --workload cm_m5 --cm-blocks $NB --cm-block-bytes 65536 --no-cpu-baseline --no-verify --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cm_m5 blocks $NB (no counters):', 'in MB/s', d['input_MBps'], 'KB/s per block', d['input_KBps_per_block'], 'ms', d['ms_per_step'])" | tee -a gpurun_out/${T}_cm_occupancy.txt
done
echo "[$(( $(date +%s) - S0 )) s] done"
