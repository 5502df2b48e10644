# Headline step under fragmenter / SHA-1 settings (run on the GPU box): serial steps give clean per-kernel times,
# then the pipelined figure for the candidates.  Usage: bash tools/frag_budget_probe.sh > gpurun_out/x.log
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
run() {  # label, extra bench args..., env comes from the caller
  L=$1; shift
  timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=d.get('kernels_ms_per_step',{})
print('$L', 'ms_per_step', d['ms_per_step'], 'serial', d.get('ms_per_step_serial'), {k:round(v,1) for k,v in ks.items() if 'frag' in k or 'sha1' in k})
"
}
S="--pipeline 1 --steps 3 --warmup 1"
ZPQ_SHA1_STAGED=0 run "A direct  budget=256K serial" $S
ZPQ_SHA1_STAGED=1 run "B staged  budget=256K serial" $S
ZPQ_SHA1_STAGED=1 ZPQ_FRAG_BUDGET=131072 run "C staged  budget=128K serial" $S
ZPQ_SHA1_STAGED=1 ZPQ_FRAG_BUDGET=98304 run "D staged  budget=96K  serial" $S
ZPQ_SHA1_STAGED=1 ZPQ_FRAG_BUDGET=65536 run "E staged  budget=64K  serial" $S
ZPQ_SHA1_STAGED=1 run "B staged  budget=256K pipelined" --steps 12 --warmup 3
ZPQ_SHA1_STAGED=1 ZPQ_FRAG_BUDGET=98304 run "D staged  budget=96K  pipelined" --steps 12 --warmup 3
ZPQ_SHA1_STAGED=1 ZPQ_FRAG_BUDGET=65536 run "E staged  budget=64K  pipelined" --steps 12 --warmup 3
