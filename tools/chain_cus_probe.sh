# Headline step (six jobs in flight) with the serial chains on compute units of their own (ZPQ_CHAIN_CUS); GPU box.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_SHA1_STAGED=${ZPQ_SHA1_STAGED:-1}
run() {
  L=$1; shift
  timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify "$@" 2>gpurun_out/probe.err | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    ks=d.get('kernels_ms_per_step',{})
    print('$L', 'ms_per_step', d['ms_per_step'], 'serial', d.get('ms_per_step_serial'), 'MB/s', d['value'], {k[:22]:round(v,1) for k,v in list(ks.items())[:9]})
except Exception as e:
    print('$L', 'FAILED', t[-300:], open('gpurun_out/probe.err').read()[-600:])
"
}
for n in ${PROBE_CUS:-0 16 32 64}; do
  ZPQ_CHAIN_CUS=$n run "chain_cus=$n pipelined" --steps 12 --warmup 3 ${PROBE_ARGS:-}
done
