# SQ counters of the method-2 path (four 64 MiB blocks): two passes of 8 counters, summed per kernel into profiles/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
M2="--workload text_m2 --text-bytes 268435456 --pipeline 1 --no-cpu-baseline --no-verify --steps 1 --warmup 0"
rm -rf $R/gpurun_out/pmc_m2_a $R/gpurun_out/pmc_m2_b
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_m2_a -o r1 -- python $R/bench.py $M2 > /dev/null 2> $R/gpurun_out/pmc_m2_a.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_m2_b -o r1 -- python $R/bench.py $M2 > /dev/null 2> $R/gpurun_out/pmc_m2_b.err
cd $R
python - <<'PY'
import glob, json, sqlite3
out = {}
for d in ("gpurun_out/pmc_m2_a", "gpurun_out/pmc_m2_b"):
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        for name, ctr, n, s in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k.startswith("rocprim"):
                k = "rocprim::" + ("radix_sort" if "radix" in name else "scan" if "scan" in name else "other")
            if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "cuda" in k:
                continue
            e = out.setdefault(k, {}).setdefault(ctr, {"launches": 0, "sum": 0.0})
            e["launches"] += n; e["sum"] += s
json.dump({"source": "rocprofv3 --pmc (two passes of 8 counters) -- python bench.py --workload text_m2 --text-bytes 268435456 --pipeline 1 --steps 1 --warmup 0 (tools/gpu_pmc_m2.sh); sums over the launches",
           "note": "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)", "kernels": out}, open("gpurun_out/r02c_pmc_sq_text_m2.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", {}).get("sum", 0))[:8]:
    g = lambda c: v.get(c, {}).get("sum", 0)
    print(k, "valu", g("SQ_INSTS_VALU"), "active_valu", g("SQ_ACTIVE_INST_VALU"), "busy", g("SQ_BUSY_CYCLES"), "gui", g("GRBM_GUI_ACTIVE"), "lds", g("SQ_INSTS_LDS"), "wait_lds", g("SQ_WAIT_INST_LDS"))
PY
rm -rf gpurun_out/pmc_m2_a gpurun_out/pmc_m2_b
