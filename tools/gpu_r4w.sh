# Round 4: who gets issue priority where waves share a SIMD -- the block checksum chains (default now) or the LZ77 parse (ZPQ_PRIO=lz)
R=$GRAFT_REPO_ROOT
cd $R
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_PLAIN=1
for e in chain lz chain lz; do
  ZPQ_PRIO=$e timeout 200 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('prio $e', d['value'], d['ms_per_step'], d['ms_per_step_serial'], {a:k[a] for a in list(k)[:4]})" | tee -a gpurun_out/r04w_prio.txt
done
