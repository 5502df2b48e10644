# Round 2: text_m2 (config 3) first measurement + kernel trace
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python bench.py --workload text_m2 --steps 1 --warmup 1 > gpurun_out/r2j_text.json 2> gpurun_out/r2j_text.err; echo "rc=$?"
tail -c 3000 gpurun_out/r2j_text.json; tail -5 gpurun_out/r2j_text.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2j_prof -o text -- python $R/bench.py --workload text_m2 --text-bytes 268435456 --steps 1 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $R/gpurun_out/r2j_prof.err
cd $R
python - <<'PY'
import glob, csv
for f in glob.glob("gpurun_out/r2j_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(r.get("Name","")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"))
PY
