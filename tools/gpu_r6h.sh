# Round 6: the candidate-traffic upper bound of config 4, MEASURED ("measure it, don't argue it").
# tools/_variants/oracle = the tree built with -DZPQ_LZ_ORACLE: the first lz77_direct4 launch records the positions the greedy chain
# decides at, the later launches request candidate lines only for those -- the producer untouched.  Streams must stay identical
# (the run verifies every block against the oracle).
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T=${1:-r06h}
S0=$(date +%s)
run() { timeout 600 python $2 bench.py --workload dup8_m1 --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/${T}_$1.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('$1:', d['value'], 'MB/s', d['ms_per_step'], 'ms; lz77_direct_kernel', k.get('lz77_direct_kernel'), 'ms per launch;', {x:v for x,v in d.items() if x.startswith('verified')})" | tee -a gpurun_out/${T}_dup8_oracle.txt; grep "lz oracle" gpurun_out/${T}_$1.err | tail -4 | tee -a gpurun_out/${T}_dup8_oracle.txt; }
if [ -z "$ZPQ_R6H_PMC_ONLY" ]; then
run tree ""
run oracle_visited_positions_only "tools/run_variant.py oracle"
fi
echo "[$(( $(date +%s) - S0 )) s] timing"
# the traffic of the oracle's second launch
cd /tmp
rm -rf /tmp/pf /tmp/pw
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r1 -- python $GRAFT_REPO_ROOT/tools/run_variant.py oracle bench.py --workload dup8_m1 --no-cpu-baseline --no-verify --steps 1 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o r1 -- python $GRAFT_REPO_ROOT/tools/run_variant.py oracle bench.py --workload dup8_m1 --no-cpu-baseline --no-verify --steps 1 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_write.err
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee -a gpurun_out/r06h_dup8_oracle.txt
import glob, sqlite3
for d, c in (("/tmp/pf", "FETCH_SIZE"), ("/tmp/pw", "WRITE_SIZE")):
    f = glob.glob(d + "/**/*_results.db", recursive=True)
    if not f:
        print(c, "no database"); continue
    cur = sqlite3.connect(f[0]).cursor()
    rows = list(cur.execute("select value from counters_collection where counter_name=? and kernel_name like '%lz77_direct4%'", (c,)))
    print(c, "KiB per lz77_direct4 launch, in launch order (first = recording pass):", [round(r[0]) for r in rows])
PY
echo "[$(( $(date +%s) - S0 )) s] done"
