# Round 6: does the spacing of the job starts matter (convoys)?  20-step windows, interleaved
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06q}
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload silesia_x256_m1 --no-cpu-baseline --no-verify --steps 20 --warmup 5 $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get('completions_s') or []
r=[round((c[i+10]-c[i])*100,1) for i in range(0,len(c)-10,5)]
print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'ten-job windows', r)" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2 3; do
sw "spacing single/depth (35 ms)" "X=1" "" | tee -a gpurun_out/${T}_sweep.txt
sw "spacing 90 ms" "ZPQ_BENCH_SPACING_MS=90" "" | tee -a gpurun_out/${T}_sweep.txt
sw "spacing 0" "ZPQ_BENCH_SPACING_MS=0.001" "" | tee -a gpurun_out/${T}_sweep.txt
done
