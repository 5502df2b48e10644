cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for e in "X=1" "ZPQ_CM_PRE_LATE=1"; do
  echo "== $e"
  env $e timeout 250 python tools/cm_perf.py 54 2048 20000 2>&1 | tail -3
  env $e timeout 120 python -m pytest tests/test_gpu_cm_spec.py -m gpu -q -x -p no:cacheprovider -k "not fixture" 2>&1 | tail -1
done
