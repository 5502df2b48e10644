# Round 5, ninth GPU call: the second-context LZ77 finder (lz77_generic_kernel) on the chip -- its parity tests against the real LZBuffer,
# the encoder subset around it, and its speed on one 16 MiB block and on 64 blocks side by side.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05i}
S0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_m3.py -m gpu -k "second_context or unsupported or lz77 or compress_block or many_blocks or methods" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_lz.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_lz.log; tail -3 gpurun_out/${T}_tests_lz.log
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/${T}_second_context_speed.txt
import sys, time
sys.path.insert(0, "tests")
import datagen, orc
from zpaqfranz_amd import Engine
e = Engine(0)
base = b"".join(b for _, b in datagen.silesia_like(seed=0))
blk = base[:(1 << 24) - 4096]
for args in ([4, 1, 4, 8, 3, 24, 1], [4, 2, 4, 8, 3, 22, 1]):
    e.lz77_encode([blk[:1 << 20]], [args])
    t = time.time(); out = e.lz77_encode([blk], [args]); dt = time.time() - t
    ok = orc.ref_lzbuffer(blk, args) == out[0] if orc.have_ref() else None
    print("args", args, "one 16 MiB block: %.2f s (%.2f MB/s), %d -> %d bytes, equal to the real LZBuffer: %s" % (dt, len(blk) / 1e6 / dt, len(blk), len(out[0]), ok))
    many = [base[(k << 21):(k << 21) + (1 << 21)] for k in range(64)]
    t = time.time(); outs = e.lz77_encode(many, [args] * 64); dt = time.time() - t
    print("   64 blocks of 2 MiB side by side: %.2f s (%.1f MB/s)" % (dt, 64 * (1 << 21) / 1e6 / dt))
PY
echo "[$(( $(date +%s) - S0 )) s] done"
