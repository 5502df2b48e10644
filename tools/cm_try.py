#!/usr/bin/env python3
"""Diagnostic: one small block through the specialised coder for a named test configuration or method."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import cmconfigs, datagen, orc
from zpaqfranz_amd import Engine, engine
name, size = sys.argv[1], int(sys.argv[2])
if name in cmconfigs.ALL:
    h = engine.compile_config(cmconfigs.ALL[name], [0] * 9)[0]
else:
    src, args = engine.make_config(engine.expand_method(name, b"x" * 1000))
    h = engine.compile_config(src, args)[0]
x = b"\0" + datagen.text_like(size, 5)
e = Engine(0)
print("engine up", name, "n =", h[6], flush=True)
(st, got), = e.cm_code([h], [x], [len(x) * 2 + 64], encode=True)
want = orc.ref_cm_encode(h, x)
print(name, "status", st, "len", len(got), "equal to reference:", got == want, flush=True)
(st, back), = e.cm_code([h], [want], [len(x) + 16], encode=False)
print(name, "decode status", st, "equal:", back == x, flush=True)
