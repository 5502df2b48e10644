# Round 6: jobs in flight again after the scratch arenas stopped over-allocating by a quarter (a job's LZ77 table states: 15.6 -> 12.8 GB)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 ZPQ_BENCH_NO_VARIANT=1 TMPDIR=/tmp
T=${1:-r06u}
sw() { local out; out=$(env $2 timeout 250 python bench.py --workload $4 --no-cpu-baseline --no-verify $3 2>gpurun_out/${T}_last.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'ms', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'depth', d.get('steps_in_flight'), 'single', (d.get('single_job') or {}).get('ms'))" 2>&1 | tail -1); echo "$1 | $2 | $out"; }
for i in 1 2; do
sw "11 in flight" "X=1" "--steps 48 --warmup 5 --pipeline 11" silesia_x256_m1 | tee -a gpurun_out/${T}_sweep.txt
sw "12 in flight" "X=1" "--steps 48 --warmup 5 --pipeline 12" silesia_x256_m1 | tee -a gpurun_out/${T}_sweep.txt
sw "13 in flight" "X=1" "--steps 48 --warmup 5 --pipeline 13" silesia_x256_m1 | tee -a gpurun_out/${T}_sweep.txt
done
sw "14 in flight" "X=1" "--steps 48 --warmup 5 --pipeline 14" silesia_x256_m1 | tee -a gpurun_out/${T}_sweep.txt
sw "text_m2, 3 in flight" "X=1" "" text_m2 | tee -a gpurun_out/${T}_sweep.txt
sw "text_m2, 4 in flight" "X=1" "--pipeline 4" text_m2 | tee -a gpurun_out/${T}_sweep.txt
sw "dup8_m1" "X=1" "" dup8_m1 | tee -a gpurun_out/${T}_sweep.txt
tail -2 gpurun_out/${T}_last.err
