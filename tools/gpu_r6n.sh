# Round 6: extract_m1 as the default bench runs it (verification, BLAKE3 and fold variants, CPU baseline) with five jobs in flight, twice
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06n}
for i in 1 2; do
timeout 600 python bench.py --workload extract_m1 2> gpurun_out/${T}_extract_$i.err | tail -1 > gpurun_out/${T}_bench_extract_$i.json; echo "rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_extract_$i.json').read()); print('extract_m1:', d['value'], d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'steps', d['steps'], 'depth', d['steps_in_flight'], 'single', (d.get('single_job') or {}).get('ms'), 'fold on', (d.get('twin_fold_on') or {}).get('ms_per_step'), 'blake3', (d.get('blake3_verify') or {}).get('ms_per_step'), 'verified', d.get('verified_all_files'), d.get('sha256_mismatches'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" | tee -a gpurun_out/${T}_extract.txt
tail -2 gpurun_out/${T}_extract_$i.err
done
