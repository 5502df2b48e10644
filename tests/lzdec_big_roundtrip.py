"""Helper of tests/test_gpu_lzdec.py (run as its own process on the GPU box): method-2 round trip of 11 blocks of 64 MiB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from zpaqfranz_amd import Engine, engine as E

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = Engine(0)
blocks = bench.text_blocks_dev(dev, 11 * ((1 << 26) - 4096), 5)
nb = len(blocks)
caps = [(eng.block_bound(n, b"", b"") + 63) & ~63 for _, n in blocks]
outs = torch.zeros(sum(caps), dtype=torch.uint8, device=dev)
jobs = (E.BlockJob * nb)()
p = 0
for k, (t, n) in enumerate(blocks):
    jobs[k].in_ = t.data_ptr(); jobs[k].n = n; jobs[k].method = b"2"; jobs[k].filename = b""; jobs[k].comment = b""; jobs[k].dosha1 = 1
    jobs[k].out = outs.data_ptr() + p; jobs[k].out_cap = caps[k]; p += caps[k]
torch.cuda.synchronize()
eng.compress_blocks_dev(jobs, nb)
assert sum(jobs[k].out_len for k in range(nb)) > (1 << 28)
for serial in ("1", "0"):
    os.environ["ZPQ_LZDEC_SERIAL"] = serial
    uj = (E.UnblockJob * nb)()
    back = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _, n in blocks]
    p = 0
    for k in range(nb):
        uj[k].in_ = outs.data_ptr() + p; uj[k].n = jobs[k].out_len; uj[k].out = back[k].data_ptr(); uj[k].out_cap = blocks[k][1] + 64; p += caps[k]
    torch.cuda.synchronize()
    eng.decompress_blocks_dev(uj, nb, True)
    torch.cuda.synchronize()
    for k in range(nb):
        n = blocks[k][1]
        assert uj[k].status == 0 and uj[k].out_len == n and bool(torch.equal(back[k][:n], blocks[k][0][:n])), (serial, k, uj[k].status)
    print("roundtrip ok serial=%s" % serial)
    del back
eng.close()
