#!/usr/bin/env python3
"""Throughput of the context-mixing coder: N blocks of S bytes under one method's model, encode then decode.
   python tools/cm_perf.py METHOD NBLOCKS BYTES [--generic]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import datagen  # noqa: E402
from zpaqfranz_amd import Engine, engine  # noqa: E402

method, nb, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if "--generic" in sys.argv:
    os.environ["ZPQ_CM_GENERIC"] = "1"
data = [b"\0" + (datagen.text_like if k % 2 else datagen.mixed)(size, 7 + k % 13) for k in range(nb)]
src, args = engine.make_config(engine.expand_method(method, data[0]))
h = engine.compile_config(src, args)[0]
e = Engine(0)
print('engine up', flush=True)
e.profile(True)
t = time.time()
enc = e.cm_code([h] * nb, data, [len(x) + 4096 for x in data], encode=True)
t1 = time.time() - t
print("encode report:", e.profile_report())
assert all(s == 0 for s, _ in enc)
t = time.time()
dec = e.cm_code([h] * nb, [g for _, g in enc], [len(x) + 16 for x in data], encode=False)
t2 = time.time() - t
print("decode report:", e.profile_report())
assert all(s == 0 and b == x for (s, b), x in zip(dec, data))
tot = sum(len(x) for x in data)
print("method %s n=%d components, %d blocks x %d B: encode %.3f s (%.3f MB/s), decode %.3f s (%.3f MB/s), ratio %.3f" %
      (method, h[6], nb, size, t1, tot / t1 / 1e6, t2, tot / t2 / 1e6, sum(len(g) for _, g in enc) / tot))
e.close()
