cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 2 3 4 6; do timeout 200 python bench.py --pipeline $d --no-cpu-baseline > gpurun_out/b_p$d.json 2> gpurun_out/b_p$d.err; done
