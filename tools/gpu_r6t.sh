# Round 6: cm_m5 with as many blocks resident as HBM holds (scratch slack capped at 256 MiB)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06t}
ZPQ_CM_TRACE=1 timeout 600 python bench.py --workload cm_m5 2> gpurun_out/${T}_cm.err | tail -1 > gpurun_out/${T}_bench_cm_m5.json; echo "rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_cm_m5.json').read()); print('cm_m5:', 'out MB/s', d['value'], 'in MB/s', d['input_MBps'], 'blocks', d['config']['blocks'], 'waves/SIMD', d['waves_per_simd'], 'KB/s per block', d['input_KBps_per_block'], 'ms', d['ms_per_step'], {k:v for k,v in d.items() if k.startswith('verified')}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'roofline', d['roofline']['frac'], d['roofline']['traffic'])" | tee gpurun_out/${T}_cm.txt
grep "cm trace.*blocks (" gpurun_out/${T}_cm.err | tail -3 | tee -a gpurun_out/${T}_cm.txt
tail -3 gpurun_out/${T}_cm.err
timeout 300 python -m pytest tests/test_gpu_cm_spec.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "many_blocks or cm_encode or methods_3_4_5 or lz77_full" 2>&1 | tail -2
