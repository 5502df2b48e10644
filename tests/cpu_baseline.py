#!/usr/bin/env python3
"""CPU baseline leg of bench.py, run as its OWN process (no torch / HIP in it): the same job on the host cores, timed
as a whole, with the REAL reference code compiled in place (oracle/_ref/libzpaqref.so: libzpaq::SHA1, LZBuffer,
Decompresser) wherever the reference tree has it; the fragment loop (missing zpaqfranz.cpp) is the restated one of
the oracle.  One worker thread per usable core -- `zpaqfranz -tN` -- the C calls release the GIL.

  cpu_baseline.py add <corpus file> <copies>                 Silesia x copies: every file fragmented + hashed, every
                                                             unique 16 MiB block compressed ("14"); whole job timed
  cpu_baseline.py dup8 <pool file> <unique units> <dup>      the dup8 workload on a bounded sample of `unique units`
                                                             16 MiB units (same duplication factor, same generator)
  cpu_baseline.py m2 <text file> <framed file> <block bytes> <framed lengths json>
                                                             method 2 ("x6,1,4,0,7,27,1"): divsufsort + LZBuffer + SHA1 of
                                                             every block; the code streams are compared with the ones
                                                             inside the GPU's framed blocks (outside the timed part)
  cpu_baseline.py cm <text file> <framed file> <block bytes> <framed lengths json> <header hex> <blocks of the full job>
                                                             context mixing: the reference Predictor (x86 JIT) + mirrored
                                                             Encoder + SHA1 over a bounded sample of the job's blocks; code
                                                             streams compared with the GPU's (outside the timed part)
  cpu_baseline.py extract <blocks file> <index file>         d blocks decoded, fragments verified (SHA-1), files
                                                             assembled and hashed (SHA-256)
Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import orc

BLOCK_LIMIT = (1 << 24) - 4096
ARGS14 = (C.c_int * 9)(4, 1, 5, 0, 3, 24, 0, 0, 0)


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


def lib():
    """(library, kind): the real reference when it has been built, else the repo's port."""
    if orc.have_ref():
        return orc._R, "reference"
    return orc._L, "port"


def frag_hash(mem, off, n):
    R, kind = lib()
    x = C.c_uint64(0)
    if kind == "reference":
        return R.ref_fragment_sha1(C.byref(mem, off), C.c_long(n), 6, C.c_uint32(4096), C.c_uint32(520192), C.byref(x))
    return orc.fragment_and_hash_view(mem, off, n)[0]


def block_cost(mem, off, n):
    R, kind = lib()
    if kind == "reference":
        return R.ref_lz1_block_cost(C.byref(mem, off), C.c_long(n), ARGS14)
    return orc.compress_block_view(mem, off, n)


def add_job(mem, files, blocks, cores):
    """files: [(off, len)] every file of the job (duplicates included); blocks: [(off, len)] the unique d blocks."""
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda t: frag_hash(mem, t[0], min(t[1], 1 << 20)), files[:cores]))          # warm the threads
        t0 = time.time()
        nfrag = sum(ex.map(lambda t: frag_hash(mem, t[0], t[1]), files))
        t1 = time.time()
        out = sum(ex.map(lambda t: block_cost(mem, t[0], t[1]), blocks))
        t2 = time.time()
    return nfrag, out, t1 - t0, t2 - t1


def main():
    mode = sys.argv[1]
    cores = usable_cores()
    _, kind = lib()
    if mode == "add":
        path, copies = sys.argv[2], int(sys.argv[3])
        sizes = json.loads(sys.argv[4]) if len(sys.argv) > 4 else None
        blob = bytearray(open(path, "rb").read())
        unit = len(blob)
        mem = (C.c_ubyte * unit).from_buffer(blob)
        if not sizes:
            sizes = [unit]
        members, o = [], 0
        for s in sizes:
            members.append((o, s)); o += s
        files = members * copies                                   # the x`copies` corpus re-reads the same bytes
        blocks = [(b, min(BLOCK_LIMIT, unit - b)) for b in range(0, unit, BLOCK_LIMIT)]
        nfrag, out, t_fh, t_c = add_job(mem, files, blocks, cores)
        total_in = unit * copies
        res = {"value": round(out / 1e6 / (t_fh + t_c), 3), "unit": "MB/s compressed output", "cores": cores, "kind": kind,
               "input_GBps": round(total_in / 1e9 / (t_fh + t_c), 4), "seconds": round(t_fh + t_c, 2),
               "sample": "WHOLE job timed, %d threads: fragment loop + libzpaq::SHA1 over %d files / %.1f GB in %.1f s (%.2f GB/s), then "
                         "LZBuffer + SHA1 of the %d unique 16 MiB blocks in %.1f s (%d fragments, code streams %.1f MB; the %d copies "
                         "re-read one %d MB copy, so the host caches help the CPU here)"
                         % (cores, len(files), total_in / 1e9, t_fh, total_in / 1e9 / t_fh, len(blocks), t_c, nfrag, out / 1e6, copies, unit >> 20)}
    elif mode == "dup8":
        path, units, dup = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        full_units = int(sys.argv[5]) if len(sys.argv) > 5 else units
        import numpy as np
        pool = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        npool = len(pool) >> 24
        blob = np.empty(units << 24, dtype=np.uint8)
        for k in range(units):                                     # same generator as bench.py: byte-rotated pool units
            np.add(pool[(k % npool) << 24:((k % npool) + 1) << 24], np.uint8(k // npool), out=blob[k << 24:(k + 1) << 24])
        mem = (C.c_ubyte * len(blob)).from_buffer(blob)
        files = [(k << 24, 1 << 24) for k in range(units)] * dup    # per-file state is reset at every unit boundary here (64 MiB files of 4 units on the GPU)
        blocks = [(b, min(BLOCK_LIMIT, len(blob) - b)) for b in range(0, len(blob), BLOCK_LIMIT)]
        nfrag, out, t_fh, t_c = add_job(mem, files, blocks, cores)
        total_in = len(blob) * dup
        res = {"value": round(out / 1e6 / (t_fh + t_c), 3), "unit": "MB/s compressed output", "cores": cores, "kind": kind,
               "input_GBps": round(total_in / 1e9 / (t_fh + t_c), 4), "seconds": round(t_fh + t_c, 2),
               "sample": "bounded sample = %d of the workload's %d unique 16 MiB units, same x%d duplication, %d threads: fragment loop + "
                         "libzpaq::SHA1 over %.1f GB in %.1f s, LZBuffer + SHA1 of %d blocks in %.1f s (rates scale with the unit count)"
                         % (units, full_units, dup, cores, total_in / 1e9, t_fh, len(blocks), t_c)}
    elif mode == "cm":
        text = open(sys.argv[2], "rb").read()
        framed = open(sys.argv[3], "rb").read()
        bs = int(sys.argv[4]); flens = json.loads(sys.argv[5]); header = bytes.fromhex(sys.argv[6]); full = int(sys.argv[7])
        nb = len(flens)
        if kind != "reference":
            raise SystemExit("cm baseline needs oracle/_ref (the real Predictor)")

        def one(k):
            x = text[k * bs: (k + 1) * bs]
            orc.ref_sha1(x)
            return orc.ref_cm_encode(header, b"\0" + x)
        with ThreadPoolExecutor(cores) as ex:
            one(0)                                                   # tables built, JIT emitted
            t0 = time.time()
            streams = list(ex.map(one, range(nb)))
            tot = time.time() - t0
        same, o = 0, 0
        for k in range(nb):
            fr = framed[o: o + flens[k]]; o += flens[k]
            kz = fr.index(b"zPQ"); p = kz + 5
            p += 2 + (fr[p] | fr[p + 1] << 8)                       # header
            p += 1                                                   # segment marker
            p = fr.index(b"\0", p) + 1; p = fr.index(b"\0", p) + 1; p += 1   # filename, comment, reserved
            same += fr[p: len(fr) - 22] == streams[k]
        out = sum(len(x) for x in streams)
        res = {"value": round(out / 1e6 / tot, 3), "unit": "MB/s compressed output", "cores": cores, "kind": kind,
               "input_MBps": round(len(text) / 1e6 / tot, 3), "seconds": round(tot, 2), "identical_blocks": "%d of %d" % (same, nb),
               "sample": "bounded sample = the first %d of the job's %d blocks (%d KiB each), %d threads: reference Predictor (x86 JIT build of "
                         "ZSFX/libzpaq.cpp) + mirrored Encoder + libzpaq::SHA1 per block, %.1f MB in %.1f s"
                         % (nb, full, bs >> 10, cores, len(text) / 1e6, tot)}
    elif mode == "m2":
        text = open(sys.argv[2], "rb").read()
        framed = open(sys.argv[3], "rb").read()
        bs = int(sys.argv[4]); flens = json.loads(sys.argv[5])
        nb = (len(text) + bs - 1) // bs
        a0 = 6
        args = (C.c_int * 9)(a0, 1, 4, 0, 7, 21 + a0, 1, 0, 0)
        R, _ = lib()
        if kind != "reference":
            raise SystemExit("m2 baseline needs oracle/_ref (the real LZBuffer)")
        mem = (C.c_ubyte * len(text)).from_buffer_copy(text)

        def one(k):
            off = k * bs; n = min(bs, len(text) - off)
            cap = n + n // 8 + 1024
            out = (C.c_ubyte * cap)()
            r = R.ref_lzbuffer(C.byref(mem, off), C.c_long(n), args, out, C.c_long(cap))
            d = (C.c_ubyte * 20)()
            R.ref_sha1(C.byref(mem, off), C.c_long(n), d)
            return bytes(out[:r])
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.time()
            streams = list(ex.map(one, range(nb)))
            tot = time.time() - t0

        def lz_stream(fr):
            """the LZ77 code stream inside a framed n=0 block: tag, zPQ, header, segment, stored sub-blocks, PCOMP preamble"""
            p = 13 + 5
            p += 2 + (fr[p] | fr[p + 1] << 8)
            assert fr[p] == 1; p += 1
            p = fr.index(b"\0", p) + 1; p = fr.index(b"\0", p) + 1
            assert fr[p] == 0; p += 1
            parts = []
            while True:
                k = int.from_bytes(fr[p:p + 4], "big"); p += 4
                if not k:
                    break
                parts.append(fr[p:p + k]); p += k
            pay = b"".join(parts)
            assert pay[0] == 1
            return pay[3 + (pay[1] | pay[2] << 8):]
        same, q = 0, 0
        for k in range(nb):
            same += lz_stream(framed[q:q + flens[k]]) == streams[k]; q += flens[k]
        out = sum(len(x) for x in streams)
        res = {"value": round(out / 1e6 / tot, 3), "unit": "MB/s compressed output", "cores": cores, "kind": kind,
               "input_GBps": round(len(text) / 1e9 / tot, 4), "seconds": round(tot, 2), "identical_blocks": "%d of %d" % (same, nb),
               "sample": "WHOLE job timed, %d threads: divsufsort + LZBuffer::fill + libzpaq::SHA1 of %d blocks of %d MiB (%.2f GB) in %.1f s"
                         % (cores, nb, bs >> 20, len(text) / 1e9, tot)}
    elif mode == "extract":
        import hashlib
        import numpy as np
        blocks_path, index_path = sys.argv[2], sys.argv[3]
        arc = open(blocks_path, "rb").read()
        ix = json.load(open(index_path))
        R, _ = lib()
        boffs, usizes = ix["block_off"], ix["block_usize"]

        def dec(k):
            f = arc[boffs[k]:boffs[k + 1]]
            if kind == "reference":
                return orc.ref_decompress_block(f, usizes[k] + 64)["data"]
            return orc.decompress_block(f, usizes[k] + 64)[0]
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.time()
            plain = list(ex.map(dec, range(len(usizes))))
            t1 = time.time()
            # fragment checksums (decompressThread, ZSFX/zsfx.cpp:1811-1834)
            frag_block, frag_off, frag_len = ix["frag_block"], ix["frag_off"], ix["frag_len"]
            sha = orc.ref_sha1 if kind == "reference" else orc.sha1
            list(ex.map(lambda i: sha(plain[frag_block[i]][frag_off[i]:frag_off[i] + frag_len[i]]), range(len(frag_len))))
            t2 = time.time()
            # files: concatenate the fragments each one points to, SHA-256 (hashlib: OpenSSL, SHA-NI where the CPU has it --
            # a generous stand-in for zpaqfranz's HWSHA2 build)
            members = ix["members"]          # [[fragment indices]] of the distinct files; every copy repeats them
            copies = ix["copies"]

            def one(m):
                h = hashlib.sha256()
                n = 0
                for i in m:
                    b = plain[frag_block[i]][frag_off[i]:frag_off[i] + frag_len[i]]
                    h.update(b); n += len(b)
                return n
            out_bytes = sum(ex.map(one, members * copies))
            t3 = time.time()
        tot = t3 - t0
        res = {"value": round(len(arc) / 1e6 / tot, 3), "unit": "MB/s compressed input", "cores": cores, "kind": kind,
               "output_GBps": round(out_bytes / 1e9 / tot, 4), "seconds": round(tot, 2),
               "sample": "WHOLE job timed, %d threads: Decompresser over %d d blocks %.2f s, libzpaq::SHA1 of %d fragments %.2f s, assembling + "
                         "SHA-256 (hashlib/OpenSSL) of %d files / %.1f GB %.2f s"
                         % (cores, len(usizes), t1 - t0, len(frag_len), t2 - t1, len(members) * copies, out_bytes / 1e9, t3 - t2)}
    else:
        raise SystemExit("unknown mode")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
