# Round 2, third GPU pass: decoder address spaces, A/B of the pipelined add against the round-1 tree, PMC counters.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 400 python -m pytest tests/test_gpu_round2.py -q -x -p no:cacheprovider -k "lz77 or resident or gather or jidac" > gpurun_out/r2c_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c_new.log
timeout 300 python bench.py --workload extract_m1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2c_extract.json 2> gpurun_out/r2c_extract.err
for i in 1 2; do
  (cd .ab_r1 && timeout 300 python bench.py --no-cpu-baseline --no-verify > $R/gpurun_out/r2c_ab_old$i.json 2> /dev/null)
  timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2c_ab_new$i.json 2> /dev/null
done
ZPQ_SHA_NO_ORDER=1 timeout 300 python bench.py --no-cpu-baseline --no-verify --pipeline 1 --steps 3 --warmup 1 > gpurun_out/r2c_p1_noorder.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r2c_counters.txt 2>&1
B="python $R/bench.py --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-kernel-timing"
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc1 -o p -- $B > /dev/null 2> $R/gpurun_out/r2c_pmc1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc2 -o p -- $B > /dev/null 2> $R/gpurun_out/r2c_pmc2.err
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc3 -o p -- $B > /dev/null 2> $R/gpurun_out/r2c_pmc3.err
cd $R
python - <<'PY'
import csv, glob, collections, json
for d in ("pmc1","pmc2","pmc3"):
    files = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name","?").split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    out = {k: {c: v / max(1, cnt[(k, c)]) for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(out, open("gpurun_out/r2c_%s.json" % d, "w"), indent=1)
    print(d, len(files), "files", len(out), "kernels")
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
tail -2 gpurun_out/r2c_new.log
for f in r2c_extract r2c_ab_old1 r2c_ab_new1 r2c_ab_old2 r2c_ab_new2 r2c_p1_noorder; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_serial"), {k:v for k,v in list(d["kernels_ms_per_step"].items())[:8]})
except Exception as e: print("ERR", e)
PY
done
tail -2 gpurun_out/r2c_pmc1.err gpurun_out/r2c_pmc2.err gpurun_out/r2c_pmc3.err
