# Round 2, fourth GPU pass: bisect the pipelined-add regression (library vs bench.py), PMC counters of the serial step.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 300 python bench_r1_copy.py --no-cpu-baseline --no-verify > gpurun_out/r2d_newlib_oldbench.json 2> gpurun_out/r2d_newlib_oldbench.err
(cd .ab_r1 && timeout 300 python bench_new_copy.py --no-cpu-baseline --no-verify > $R/gpurun_out/r2d_oldlib_newbench.json 2> $R/gpurun_out/r2d_oldlib_newbench.err)
(cd .ab_r1 && timeout 300 python bench.py --no-cpu-baseline --no-verify > $R/gpurun_out/r2d_old.json 2> /dev/null)
timeout 300 python bench.py --no-cpu-baseline --no-verify > gpurun_out/r2d_new.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-kernel-timing"
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc1 -o p -- $B > /dev/null 2> $R/gpurun_out/r2d_pmc1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc2 -o p -- $B > /dev/null 2> $R/gpurun_out/r2d_pmc2.err
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc3 -o p -- $B > /dev/null 2> $R/gpurun_out/r2d_pmc3.err
cd $R
python - <<'PY'
import sqlite3, glob, json
for d in ("pmc1","pmc2","pmc3"):
    f = glob.glob("gpurun_out/%s/**/*_results.db" % d, recursive=True)
    if not f: print(d, "no db"); continue
    cur = sqlite3.connect(f[0]).cursor()
    out = {}
    try:
        for name, ctr, calls, sm in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            k = name.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:48]
            out.setdefault(k, {})[ctr] = [calls, sm]
    except Exception as e:
        print(d, "query failed", e, [r[0] for r in cur.execute("select name from sqlite_master")][:40])
    json.dump(out, open("gpurun_out/r2d_%s.json" % d, "w"), indent=1)
    print(d, len(out), "kernels")
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
for f in r2d_newlib_oldbench r2d_oldlib_newbench r2d_old r2d_new; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_serial"))
except Exception as e: print("ERR", e)
PY
done
tail -3 gpurun_out/r2d_newlib_oldbench.err gpurun_out/r2d_oldlib_newbench.err
