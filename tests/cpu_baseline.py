#!/usr/bin/env python3
"""CPU baseline leg of bench.py, run as its OWN process (no torch / HIP in it, so that forking workers is safe):
the oracle (oracle/liboracle.so, a port of the reference's path) on all host cores, one worker process per core
as `zpaqfranz -tN` uses threads, on a bounded sample of the corpus file given as argv[1].

Prints one JSON object.  Sample: fragment + SHA-1 of 4 x 8 MiB per core; compressBlock("14") of one 16 MiB block
per core (at most 64).  The x`copies` job is extrapolated as copies x fragment/hash + 1 x compress of one copy."""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import orc

BLOCK_LIMIT = (1 << 24) - 4096
_mem = None


def _fh(t):
    return orc.fragment_and_hash_view(_mem, t[0], t[1])[0]


def _cb(t):
    return orc.compress_block_view(_mem, t[0], t[1])


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs but grants a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p) + 0.5)))
    except Exception:
        pass
    return n


def main():
    global _mem
    path, copies = sys.argv[1], int(sys.argv[2])
    blob = bytearray(open(path, "rb").read())
    unit = len(blob)
    _mem = (C.c_ubyte * unit).from_buffer(blob)
    cores = usable_cores()
    piece = min(8 << 20, unit)
    tasks = [((i * 7919 * 4096) % max(1, unit - piece), piece) for i in range(cores * 4)]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_fh, tasks[:cores])                    # warm the workers
        t0 = time.time(); pool.map(_fh, tasks, chunksize=1); dt_fh = time.time() - t0
        nblk = max(1, min(cores, 64))
        bl = min(BLOCK_LIMIT, unit)
        ctasks = [((i * bl) % max(1, unit - bl), bl) for i in range(nblk)]
        t1 = time.time(); outs = pool.map(_cb, ctasks, chunksize=1); dt_c = time.time() - t1
    nbytes = sum(t[1] for t in tasks)
    t_fh = dt_fh / nbytes                               # s per input byte on all cores
    cin, cout = nblk * bl, sum(outs)
    nblocks_job = max(1, (unit + BLOCK_LIMIT - 1) // BLOCK_LIMIT)
    # nblk blocks ran concurrently; the real job has nblocks_job blocks for min(cores, nblocks_job) workers
    waves = -(-nblocks_job // min(cores, nblocks_job))
    t_comp_job = dt_c * waves * (1.0 if nblk >= min(cores, nblocks_job) else min(cores, nblocks_job) / nblk)
    est_time = t_fh * unit * copies + t_comp_job
    est_out = unit * (cout / max(1, cin))
    print(json.dumps({"value": round(est_out / 1e6 / est_time, 3), "unit": "MB/s compressed output", "cores": cores, "kind": "port",
                      "input_GBps": round(unit * copies / 1e9 / est_time, 4),
                      "sample": "oracle/liboracle.so, %d worker processes: fragment+SHA-1 of %d MB in %.2f s (%.1f GB/s), "
                                "compressBlock('14') of %d x 16 MiB concurrently in %.2f s (ratio %.3f); extrapolated to "
                                "%d x fragment/hash + 1 x compress of the %d MB unique copy"
                                % (cores, nbytes >> 20, dt_fh, nbytes / 1e9 / dt_fh, nblk, dt_c, cout / max(1, cin), copies, unit >> 20)}))


if __name__ == "__main__":
    main()
