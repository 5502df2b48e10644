import ctypes, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import datagen
from zpaqfranz_amd import Engine
e = Engine(0)
blocks = [datagen.mixed((1 << 24) - 4096, 70 + i) for i in range(4)]
args = [[4, 1, 5, 0, 3, 24]] * 4
e.lz77_encode(blocks, args)
out = (ctypes.c_ulonglong * 8)()
e.L.zpq_debug_lzprof(out, 1)
t = time.time(); e.lz77_encode(blocks, args); dt = time.time() - t
e.L.zpq_debug_lzprof(out, 1)
v = list(out); tot = sum(v)
print("lz77 4 x 16 MiB: %.1f ms wall; cycles by phase (hash, table+forward, candidates, decision, chain, insert, -, -):" % (dt * 1e3))
print([round(x / tot, 3) for x in v], "total cycles/window = %.0f" % (tot / (4 * (1 << 24) / 64)))
