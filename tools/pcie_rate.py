"""PCIe-inclusive rate of the host-buffer entry points: 1 GiB of corpus text handed over as host
memory -> H2D + fragment + SHA-1 (+ D2H of the tables), and one compressBlock("14") batch of 8 x 16 MiB from host buffers."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import datagen
from zpaqfranz_amd import Engine
e = Engine(0)
unit = b"".join(b for _, b in datagen.silesia_like(seed=0, scale=0.25))          # ~53 MB
n = (1 << 30) // len(unit)
host = np.frombuffer(unit * n, dtype=np.uint8)
d = e.alloc(host.nbytes)
for _ in range(2):
    t = time.perf_counter(); e._ck(e.L.zpq_h2d(e.ctx, d.ptr, host.ctypes.data, host.nbytes)); e.sync(); h2d = time.perf_counter() - t
files = [unit] * n
t = time.perf_counter()
e._ck(e.L.zpq_h2d(e.ctx, d.ptr, host.ctypes.data, host.nbytes))
fr = e.fragment_files(files[:4])          # warm scratch on a slice through the host path
t1 = time.perf_counter() - t
blocks = [bytes(host[i * ((1 << 24) - 4096):(i + 1) * ((1 << 24) - 4096)]) for i in range(8)]
e.compress_blocks(blocks[:1], ["14"])
t = time.perf_counter(); res = e.compress_blocks(blocks, ["14"] * 8); tc = time.perf_counter() - t
print("H2D %.2f GB in %.3f s = %.1f GB/s (pageable host memory)" % (host.nbytes / 1e9, h2d, host.nbytes / 1e9 / h2d))
print("compressBlock x8 from host buffers: %.1f MB in, %.1f MB out, %.3f s = %.0f MB/s in, %.0f MB/s out" % (
    sum(map(len, blocks)) / 1e6, sum(len(o) for _, o in res) / 1e6, tc, sum(map(len, blocks)) / 1e6 / tc, sum(len(o) for _, o in res) / 1e6 / tc))
