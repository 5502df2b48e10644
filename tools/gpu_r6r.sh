# Round 6: dealt d blocks (add_impl) on the chip: the sharded / multi-context / two-rank tests, then the whole suite once more
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06r}
timeout 1800 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > gpurun_out/${T}_tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_tests_gpu.log; tail -10 gpurun_out/${T}_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
ZPQJ_TRACE_OWNERS=1 ZPQ_BENCH_NO_VARIANT=1 timeout 300 python bench.py --workload silesia_x256_m1 --force-collectives --no-cpu-baseline --no-verify --steps 20 --warmup 5 2> gpurun_out/${T}_bench_rccl1.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rccl world 1:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'))"
grep "zpqj add" gpurun_out/${T}_bench_rccl1.err | sort | uniq -c | head -3
