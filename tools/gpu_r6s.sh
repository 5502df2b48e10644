# Round 6: context mixing with more blocks resident (the MI355X has 288 GiB = 309 GB: 2048 blocks x 90 MB leave room)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06s}
for NB in ${NBS:-2816 2944}; do
  ZPQ_CM_TRACE=1 timeout 400 python bench.py --workload cm_m5 --cm-blocks $NB --no-cpu-baseline --steps 1 --warmup 1 2> gpurun_out/${T}_cm_$NB.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cm_m5 blocks $NB:', 'out MB/s', d['value'], 'in MB/s', d['input_MBps'], 'waves/SIMD', d['waves_per_simd'], 'KB/s per block', d['input_KBps_per_block'], 'ms', d['ms_per_step'], {k:v for k,v in d.items() if k.startswith('verified')})" | tee -a gpurun_out/${T}_cm_blocks.txt
  grep "cm trace.*blocks (" gpurun_out/${T}_cm_$NB.err | tail -2 | tee -a gpurun_out/${T}_cm_blocks.txt
  tail -2 gpurun_out/${T}_cm_$NB.err | grep -i "error\|memory" | head -2
done
