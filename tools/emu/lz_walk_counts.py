"""How many dependent steps does the LZ77 walk take per window?  Runs the hash-table parse of a Silesia-like sample on the
emulated engine built with event counters (ZPQ_LZ_COUNT) in three forms -- table states (default), candidate tables, candidate
tables software-pipelined -- and prints events per 64-byte window.  No timing: what a window costs on the GPU is roughly
(memory round trips on its chain) x (latency); this counts the round trips' causes.
usage: python tools/emu/lz_walk_counts.py [scale]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import emu_build
emu_build.activate()
import datagen
from zpaqfranz_amd import Engine, engine
eng = Engine(0)
L = engine.load()
args = [4, 1, 5, 0, 3, 24]
tot = [0] * 8
nbytes = 0
for name, b in datagen.silesia_like(seed=3, scale=%(scale)s):
    b = b[:1 << 20]
    if len(b) < 4096:
        continue
    out = (C.c_ulonglong * 8)()
    L.zpq_debug_lzcount(out, 1)
    eng.lz77_encode([b], [args])
    L.zpq_debug_lzcount(out, 1)
    w = max(1, len(b) // 64)
    print("  %%-10s %%8d B  looked %%5.2f  insert-only %%5.2f  pipe windows %%5.2f | tokens %%5.2f  exact re-evaluations %%5.2f  extension rounds %%5.2f  wave compares %%5.2f  restarts %%5.2f"
          %% (name[:10], len(b), out[0] / w, out[1] / w, out[7] / w, out[2] / w, out[3] / w, out[4] / w, out[5] / w, out[6] / w))
    for i in range(8):
        tot[i] += out[i]
    nbytes += len(b)
w = max(1, nbytes // 64)
print("  TOTAL      %%8d B  looked %%5.2f  insert-only %%5.2f  pipe windows %%5.2f | tokens %%5.2f  exact re-evaluations %%5.2f  extension rounds %%5.2f  wave compares %%5.2f  restarts %%5.2f"
      %% (nbytes, tot[0] / w, tot[1] / w, tot[7] / w, tot[2] / w, tot[3] / w, tot[4] / w, tot[5] / w, tot[6] / w))
"""


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "0.01"
    for label, env in (("table states (default), one wave per block", {"ZPQ_LZ_DIRECT": "1"}),
                       ("candidate tables", {"ZPQ_LZ_CAND": "1", "ZPQ_LZ_DIRECT": "1"}),
                       ("candidate tables, pipelined walk", {"ZPQ_LZ_CAND": "1", "ZPQ_LZ_CAND_PIPE": "1", "ZPQ_LZ_DIRECT": "1"})):
        print(label + " (events per 64-byte window)")
        e = dict(os.environ, ZPQ_EMU_COUNT="1", **env)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "scale": scale}], env=e, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode:
            print(r.stderr[-2000:])


if __name__ == "__main__":
    main()
