cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for d in 2 3 4; do
  timeout 300 python bench.py --workload text_m2 --no-cpu-baseline --no-verify --pipeline $d --steps $((d*3)) --warmup 1 2>gpurun_out/probe.err | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('text_m2 depth $d', d['ms_per_step'], d['value'], d['unit'])
except Exception as e: print('FAILED', t[-300:], open('gpurun_out/probe.err').read()[-500:])
"
done
