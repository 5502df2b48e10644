"""Cycles per phase of the LZ77 table walks (engine built with -DZPQ_LZ_PROFILE under tools/_prof, or the directory ZPQ_PROF_DIR
names: a tools/make_variant.sh variant built with that flag): one process per setting."""
import ctypes, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r'''
import ctypes, sys, os, time
sys.path.insert(0, os.environ.get("ZPQ_PROF_DIR") or os.path.join(%(root)r, "tools", "_prof")); sys.path.insert(1, os.path.join(%(root)r, "tests"))
import datagen
from zpaqfranz_amd import Engine
e = Engine(0)
nb = int(os.environ.get("NB", "4"))
if os.environ.get("CORPUS") == "silesia":
    base = b"".join(b for _, b in datagen.silesia_like(seed=0))
    bs = int(os.environ.get("BS", "24"))
    m = len(base) >> bs
    pieces = [base[i << bs:((i + 1) << bs) - 4096] for i in range(min(m, nb))]
    blocks = [pieces[i %% len(pieces)] for i in range(nb)]
else:
    bs = 24
    blocks = [datagen.mixed((1 << 24) - 4096, 70 + i) for i in range(nb)]
args = [[4, 1, 5, 0, 3, 24]] * nb
e.lz77_encode(blocks[:12], args[:12])
out = (ctypes.c_ulonglong * 8)()
e.L.zpq_debug_lzprof(out, 1)
_o2 = (ctypes.c_ulonglong * 24)(); e.L.zpq_debug_lzprof2(_o2, 1)
t = time.time(); e.lz77_encode(blocks, args); dt = time.time() - t
e.L.zpq_debug_lzprof(out, 1)
v = list(out); tot = sum(v)
out2 = (ctypes.c_ulonglong * 24)()
e.L.zpq_debug_lzprof2(out2, 1)
w = nb * (1 << bs) / 64
print("%%s: %%.1f ms wall" %% (os.environ.get("LABEL"), dt * 1e3))
if tot:
    print("   one wave: cycles/window %%.0f; by phase (hash, rows+forward, candidates, decision, chain, insert): %%s" %% (tot / w, [round(x / w) for x in v[:6]]))
if sum(out2):
    v2 = list(out2)
    print("   producer  cycles/window (ring wait, hash, rows+forward, ring write, insert+wait, swallowed): %%s = %%.0f" %% ([round(x / w) for x in v2[:6]], sum(v2[:8]) / w))
    print("   evaluator cycles/window (ring-1 wait, read, ring-2 wait, candidates, decision, swallowed, write): %%s = %%.0f" %% ([round(x / w) for x in v2[8:15]], sum(v2[8:16]) / w))
    print("   chain     cycles/window (ring-2 wait, read, -, -, chain): %%s = %%.0f" %% ([round(x / w) for x in v2[16:21]], sum(v2[16:21]) / w))
    if v2[21] + v2[22]:
        print("   emitter   cycles/window (token wait, code stream): %%s" %% [round(x / w) for x in v2[21:23]])
''' % {"root": ROOT}
SETS = (("silesia-like units x 1091 (the chip full), one workgroup per block", {"ZPQ_LZ_DIRECT": "1", "CORPUS": "silesia", "NB": "1091"}),
        ("silesia-like units, one workgroup per block", {"ZPQ_LZ_DIRECT": "1", "CORPUS": "silesia", "NB": "12"}),
                   ("silesia-like units, spec three waves", {"CORPUS": "silesia", "NB": "12"}),
        ("mixed, spec three waves", {}))
if len(sys.argv) > 1:
    SETS = SETS[:int(sys.argv[1])]
for label, env in SETS:
    e = dict(os.environ, LABEL=label, **env)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=e, timeout=600)
    print(r.stdout.strip() or r.stderr[-800:])
