# Headline step under a list of settings (GPU box).  Each line of $1: label|ENV=.. ENV=..|bench args
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
while IFS='|' read -r L E A; do
  [ -z "$L" ] && continue
  env $E timeout 240 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify $A 2>gpurun_out/probe.err | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    ks=d.get('kernels_ms_per_step',{})
    print('$L', '| ms_per_step', d['ms_per_step'], 'serial', d.get('ms_per_step_serial'), 'MB/s', d['value'], {k[:20]:round(v) for k,v in list(ks.items())[:8]})
except Exception as e:
    print('$L', 'FAILED', t[-300:], open('gpurun_out/probe.err').read()[-600:])
"
done < $1
