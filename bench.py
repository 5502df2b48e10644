#!/usr/bin/env python3
"""bench.py -- zpaqfranz "add -m1" hot path on MI355X: fragment -> SHA-1 -> dedup -> pack -> LZ77 level 1
-> ZPAQ block framing, whole job per step, input resident in HBM.

Workload (BASELINE.json configs[1]): the Silesia corpus replicated x256 (3072 files, 54 256 276 480
bytes) at -m1 (16 MiB blocks, method "14" -> x4,1,5,0,3,24).  The real corpus cannot be fetched here, so
a seeded synthetic corpus with Silesia's member names and sizes stands in ("data": "synthetic").

One JSON line is printed by rank 0 (see the contract in the task statement).  metric/unit are
BASELINE.json's: MB/s of compressed archive output; input-side GB/s is reported next to it because
dedup collapses the x256 corpus to one copy before compression.

N > 1 (torchrun, one rank per GPU, RCCL): weak scaling -- every rank owns its own x256 corpus (different
seed), fragments and hashes it, the fragment tables are all-gathered over RCCL, every rank resolves the
global first-occurrence dedup identically, blocks are packed by the one deterministic global packer and
owned by the rank that holds their first fragment (seam fragments travel peer to peer), and the
compressed blocks are all-gathered so that rank 0 could stitch the archive in fixed block order."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from zpaqfranz_amd.sharding import BLOCK_LIMIT, plan as shard_plan
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


_CPU_COLLECTIVES = False     # set when the process group is gloo (functional test on one GPU)


def _all_gather(tensor):
    """all_gather of equally-shaped tensors -> list; over RCCL on device tensors, or via host for gloo."""
    world = dist.get_world_size()
    if _CPU_COLLECTIVES:
        t = tensor.cpu()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.to(tensor.device) for o in out]
    out = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(out, tensor)
    return out


def _exchange(sends, recvs, device):
    """sends: {dst: uint8 tensor}, recvs: {src: nbytes} -> {src: uint8 tensor}; one batched P2P group."""
    got = {src: torch.empty(n, dtype=torch.uint8, device="cpu" if _CPU_COLLECTIVES else device) for src, n in recvs.items()}
    ops = [dist.P2POp(dist.isend, (t.cpu() if _CPU_COLLECTIVES else t), int(dst)) for dst, t in sorted(sends.items())]
    ops += [dist.P2POp(dist.irecv, got[src], int(src)) for src in sorted(got)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return {src: t.to(device) for src, t in got.items()}


class CollectiveOrder:
    """With several steps in flight on several threads, every rank must still issue its collectives in ONE order.
    A step has three collective sections (0: fragment tables, 1: seam fragments, 2: compressed streams); the order is
    the software-pipeline order  (k,0) (k,1) (k-depth+1, 2)  for k = 0, 1, ...: a fixed function of (steps, depth),
    hence identical on every rank, and one that lets step k exchange its tables before step k-1 has finished
    compressing.  A section holds the turn until its transfers have completed."""

    def __init__(self, steps, depth):
        import threading
        self.seq = []
        for k in range(steps + depth):
            for st, sec in ((k, 0), (k, 1), (k - (depth - 1), 2)):
                if 0 <= st < steps:
                    self.seq.append((st, sec))
        self.head = 0
        self.cv = threading.Condition()

    def enter(self, step, sec, timeout=600.0):
        with self.cv:
            t0 = time.monotonic()
            while self.seq[self.head] != (step, sec):
                self.cv.wait(5.0)
                if time.monotonic() - t0 > timeout:       # fail loudly rather than hang the job
                    raise RuntimeError("collective order stalled at %r waiting for %r" % (self.seq[self.head], (step, sec)))

    def leave(self):
        with self.cv:
            self.head += 1
            self.cv.notify_all()


class _NoOrder:          # single-threaded use (sizing steps, one rank)
    def enter(self, step, sec): pass
    def leave(self): pass


class Pipeline:
    def __init__(self, eng, device, corpus, copies, rank, world, share=None):
        from zpaqfranz_amd import engine as E
        self.E, self.eng, self.dev, self.rank, self.world = E, eng, device, rank, world
        base = b"".join(b for _, b in corpus)
        sizes = [len(b) for _, b in corpus]
        self.unit = len(base)
        self.total = self.unit * copies
        if share is not None:
            self.data = share.data                       # the input is read-only: pipelines share it
        else:
            self.data = torch.empty(self.total + 64, dtype=torch.uint8, device=device)
            hb = torch.frombuffer(bytearray(base), dtype=torch.uint8)
            self.data[: self.unit].copy_(hb)
            for c in range(1, copies):
                self.data[c * self.unit:(c + 1) * self.unit].copy_(self.data[: self.unit])
            self.data[self.total:].zero_()
        off = [0]
        for c in range(copies):
            for s in sizes:
                off.append(off[-1] + s)
        self.file_off = off
        self.nfiles = len(off) - 1
        self.params = eng.fragment_params()
        self.cap = eng.fragment_capacity(off, self.params)
        i64, i32, u8 = torch.int64, torch.int32, torch.uint8
        self.frag_off = torch.empty(self.cap, dtype=i64, device=device)
        self.frag_len = torch.empty(self.cap, dtype=i32, device=device)
        self.frag_file = torch.empty(self.cap, dtype=i32, device=device)
        self.digests = torch.empty(self.cap * 20 + 64, dtype=u8, device=device)
        self.first = torch.empty(self.cap * max(1, world), dtype=i32, device=device)
        self.tstream = torch.cuda.Stream(device=device)
        torch.cuda.synchronize()

    def step(self, order=None, idx=0):
        return self.phase_b(self.phase_a(), order or _NoOrder(), idx)

    def phase_a(self):
        """Local half of a step: fragment + SHA-1 of every fragment.  No collective, so a helper thread may run it
        for step i+1 while the main thread is in phase_b of step i (the multi-rank pipeline)."""
        eng = self.eng
        with torch.cuda.stream(self.tstream):
            nf = eng.fragment_dev(self.data.data_ptr(), self.file_off, self.params, self.frag_off.data_ptr(),
                                  self.frag_len.data_ptr(), self.frag_file.data_ptr(), self.cap)
            eng.sha1_extents_dev(self.data.data_ptr(), self.frag_off.data_ptr(), self.frag_len.data_ptr(), nf,
                                 self.digests.data_ptr())
            eng.sync()
        return nf

    def phase_b(self, nf, order, idx):
        with torch.cuda.stream(self.tstream):
            return self._rest(nf, order, idx)

    def _rest(self, nf, order, idx):
        eng, E, dev = self.eng, self.E, self.dev
        tsync = self.tstream.synchronize   # waits for THIS pipeline's torch work only (another step may be in flight)
        dig, flen = self.digests[: nf * 20], self.frag_len[:nf]
        my_lo = 0
        if self.world > 1:
            # exchange: fragment tables (20-byte id + length) of every rank, order-preserving
            order.enter(idx, 0)
            cnt = torch.tensor([nf], dtype=torch.int64, device=dev)
            cnts = [int(c.item()) for c in _all_gather(cnt)]
            mx = max(cnts)
            pad_d = torch.zeros(mx * 20, dtype=torch.uint8, device=dev); pad_d[: nf * 20] = dig
            pad_l = torch.zeros(mx, dtype=torch.int32, device=dev); pad_l[:nf] = flen
            gd = _all_gather(pad_d); gl = _all_gather(pad_l)
            dig = torch.cat([gd[r][: cnts[r] * 20] for r in range(self.world)] + [torch.zeros(64, dtype=torch.uint8, device=dev)])
            flen = torch.cat([gl[r][: cnts[r]] for r in range(self.world)])
            my_lo = sum(cnts[: self.rank])
            ntot = sum(cnts)
            tsync()
            order.leave()
        else:
            cnts, ntot = [nf], nf
        # 3. dedup (global first occurrence)
        eng.dedup_dev(dig.data_ptr(), ntot, self.first.data_ptr())
        eng.sync()
        first = self.first[:ntot].cpu().numpy()
        lens = flen.cpu().numpy().astype(np.int64)
        # 4. pack (host: which unique fragment goes to which block, who owns it) -- zpaqfranz_amd/sharding.py
        P = shard_plan(first, lens, cnts, self.rank)
        uniq_idx, mine, starts, nblk = P["uniq_idx"], P["mine"], P["starts"], P["nblocks"]
        # block buffers of the blocks this rank owns: fragments + size table + 0 + count
        blk_n, src_off, src_len, dst_off, trailers, layout = [], [], [], [], [], {}
        pos = 0
        for b in mine:
            u, l, own_rank = P["blocks"][int(b)]
            o = pos + np.concatenate(([0], np.cumsum(l)[:-1]))
            own = own_rank == self.rank
            src_off.append(u[own] - my_lo); src_len.append(l[own]); dst_off.append(o[own])
            layout.update({int(g): int(d) for g, d in zip(u[~own].tolist(), o[~own].tolist())})
            size = int(l.sum())
            tr = np.concatenate((l.astype("<u4"), np.array([0, len(l)], dtype="<u4"))).tobytes()
            trailers.append((pos + size, tr))
            blk_n.append(size + len(tr))
            pos += (size + len(tr) + 63 + 64) & ~63
        blocks_buf = torch.empty(max(pos, 64), dtype=torch.uint8, device=dev)
        if mine.size:
            so = torch.from_numpy(np.concatenate(src_off).astype(np.int64)).to(dev)
            sl = torch.from_numpy(np.concatenate(src_len).astype(np.int32)).to(dev)
            do = torch.from_numpy(np.concatenate(dst_off).astype(np.int64)).to(dev)
            abs_off = self.frag_off[:nf][so]
            tsync()
            eng.gather_dev(self.data.data_ptr(), abs_off.data_ptr(), sl.data_ptr(), do.data_ptr(), so.numel(),
                           blocks_buf.data_ptr())
            for p_, tr in trailers:
                blocks_buf[p_:p_ + len(tr)] = torch.frombuffer(bytearray(tr), dtype=torch.uint8).to(dev)
        if self.world > 1:
            order.enter(idx, 1)
            self._exchange_seams(P, lens, my_lo, nf, layout, blocks_buf)
            tsync()
            order.leave()
        eng.sync(); tsync()
        # 5. compressBlock on every owned block ("14": LZ77 x4,1,5,0,3,24 + framing + SHA-1)
        nb = len(mine)
        out_bytes = 0
        outs = None
        if nb:
            jobs = (E.BlockJob * nb)()
            caps = [eng.block_bound(n, b"jDC20240101000000d0000000001", b"jDC\x01") for n in blk_n]
            ocap = [(c + 63) & ~63 for c in caps]
            outs = torch.empty(sum(ocap), dtype=torch.uint8, device=dev)
            p_in, p_out = 0, 0
            names = []
            for k, b in enumerate(mine):
                first_id = int(starts[b]) + 1   # 1-based id of the block's first (new) fragment
                names.append(("jDC20240101000000d%010d" % first_id).encode())
                jobs[k].in_ = blocks_buf.data_ptr() + p_in
                jobs[k].n = blk_n[k]
                jobs[k].method = b"14"
                jobs[k].filename = names[-1]
                jobs[k].comment = b"jDC\x01"
                jobs[k].dosha1 = 0 if getattr(self, 'no_block_sha1', False) else 1
                jobs[k].out = outs.data_ptr() + p_out
                jobs[k].out_cap = ocap[k]
                p_in += (blk_n[k] + 63 + 64) & ~63
                p_out += ocap[k]
            tsync()
            eng.compress_blocks_dev(jobs, nb)
            out_bytes = sum(jobs[k].out_len for k in range(nb))
            self.last_blocks = [(int(mine[k]), jobs[k].out_len) for k in range(nb)]
            # kept for --verify (outside the timed region): input and framed output of the first block
            self.verify_sample = (blocks_buf[: blk_n[0]], outs[: jobs[0].out_len], names[0])
            q_, pieces = 0, []
            for k in range(nb):
                pieces.append(outs[q_:q_ + jobs[k].out_len]); q_ += ocap[k]
            self.verify_sample_all = torch.cat(pieces)
        if self.world > 1:
            # the archive is stitched on rank 0 in block order: gather the compressed streams
            order.enter(idx, 2)
            t = torch.tensor([out_bytes], dtype=torch.int64, device=dev)
            ts = _all_gather(t)
            mx = max(int(x.item()) for x in ts)
            buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
            if outs is not None:
                q = 0; p_out = 0
                for k in range(nb):
                    buf[q:q + jobs[k].out_len] = outs[p_out:p_out + jobs[k].out_len]
                    q += jobs[k].out_len; p_out += ocap[k]
            gathered = _all_gather(buf)
            out_bytes = sum(int(x.item()) for x in ts)
            self.gathered = [g[: int(x.item())] for g, x in zip(gathered, ts)]   # rank r's framed blocks, block order
            tsync()
            order.leave()
        self.stats = dict(fragments=int(ntot), unique_fragments=int(len(uniq_idx)), blocks=int(nblk),
                          unique_bytes=int(lens[uniq_idx].sum()), out_bytes=int(out_bytes))
        return out_bytes

    def _exchange_seams(self, P, lens, my_lo, nf, layout, blocks_buf):
        """Fragments of a block that live on another rank (only at rank seams) travel peer to peer:
        both sides derive the same ordered lists from the plan, so one send/recv per rank pair suffices."""
        dev = self.dev
        sends, recvs = {}, {}
        for dst, idx in P["send"].items():
            loc = torch.from_numpy((idx - my_lo).astype(np.int64)).to(dev)
            offs = self.frag_off[:nf][loc].tolist()
            sends[int(dst)] = torch.cat([self.data[o:o + int(l)] for o, l in zip(offs, lens[idx].tolist())])
        for src, idx in P["recv"].items():
            recvs[int(src)] = int(lens[idx].sum())
        got = _exchange(sends, recvs, dev)
        for src, idx in P["recv"].items():
            buf, q = got[int(src)], 0
            for g, ll in zip(idx.tolist(), lens[idx].tolist()):
                d = layout[int(g)]
                blocks_buf[d:d + ll] = buf[q:q + ll]; q += ll


def cpu_baseline(corpus, copies):
    """Runs tests/cpu_baseline.py (the CPU oracle on all host cores) in a fresh process and returns its JSON."""
    import subprocess
    import tempfile
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.NamedTemporaryFile(dir=d, suffix=".corpus") as f:
        for _, b in corpus:
            f.write(b)
        f.flush()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cpu_baseline.py"), f.name, str(copies)],
                           capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"value": None, "error": r.stderr[-300:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--copies", type=int, default=256, help="corpus replication factor (256 = BASELINE config)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every corpus member (debug only)")
    ap.add_argument("--pipeline", type=int, default=3, help="steps in flight (each on its own engine context); 1 = strictly serial")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL); gloo runs the collectives through the host: functional test only")
    ap.add_argument("--same-device", action="store_true", help="test only: every rank uses GPU 0 (with --dist-backend gloo)")
    ap.add_argument("--dump-archive", default=None, help="test only: rank 0 writes the stitched d blocks of the last step to this file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block-sha1", action="store_true", help="experiment only: skip the per-block SHA-1 (invalid as a result)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with hipEvents (roofline block is then empty)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run bit-identity check against the oracle")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.same_device:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        global _CPU_COLLECTIVES
        if a.dist_backend == "gloo":
            _CPU_COLLECTIVES = True
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    import datagen
    from zpaqfranz_amd import Engine
    eng = Engine(local)
    corpus = datagen.silesia_like(seed=rank, scale=a.scale)
    # `pipeline` steps in flight on as many engine contexts and threads; with several ranks the collectives of the
    # steps in flight are issued in one fixed order on every rank (CollectiveOrder)
    depth = max(1, a.pipeline)
    pipes = [Pipeline(eng, dev, corpus, a.copies, rank, world)]
    engines = [eng]
    for _ in range(1, depth):
        e2 = Engine(local)
        engines.append(e2)
        pipes.append(Pipeline(e2, dev, corpus, a.copies, rank, world, share=pipes[0]))
    pipe = pipes[0]
    for p_ in pipes:
        p_.no_block_sha1 = a.no_block_sha1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for e_ in engines:
            e_.sync()

    def run_steps(n):
        """n whole-job steps; with depth > 1 they are software-pipelined: while one step sits in its
        latency-bound tail (LZ77 parse, block checksums: a few hundred waves) the next one already
        fragments and hashes on the rest of the chip.  Every step is complete when this returns."""
        import threading
        nxt, lock, outs, errs = [0], threading.Lock(), [0] * n, []

        order = CollectiveOrder(n, depth) if world > 1 else _NoOrder()

        def worker(p_, delay):
            try:
                torch.cuda.set_device(local)     # the current device is per host thread
                time.sleep(delay)      # stagger: one step's chip-wide kernels against the other's latency-bound tail
                while True:
                    with lock:
                        i = nxt[0]; nxt[0] += 1
                    if i >= n:
                        return
                    outs[i] = p_.step(order, i)
                    if i == n - 1:
                        last_pipe[0] = p_
            except Exception as ex:       # surface worker failures in the main thread
                errs.append(ex)
                if world > 1:              # a rank that stops would leave the others waiting in a collective
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
                    os._exit(3)
        if depth == 1:
            worker(pipes[0], 0.0)
        else:
            th = [threading.Thread(target=worker, args=(p_, k_ * stagger[0] / depth)) for k_, p_ in enumerate(pipes)]
            for t in th: t.start()
            for t in th: t.join()
        if errs:
            raise errs[0]
        return outs[-1] if n else 0

    stagger = [0.0]
    last_pipe = [pipes[0]]     # the context that ran the last step (its results are the ones dumped / verified)
    if depth > 1:            # one untimed serial step per context sizes its scratch; a second, warm one gives the stagger
        for p_ in pipes:
            p_.step()
        t_ = time.perf_counter(); pipes[0].step(); stagger[0] = time.perf_counter() - t_
    run_steps(a.warmup)
    for e_ in engines:
        e_.profile(not a.no_kernel_timing)
    barrier()
    t0 = time.perf_counter()
    out_bytes = run_steps(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    pipe = last_pipe[0]
    kern = {}
    for e_ in engines:
        for k_, (c_, m_) in e_.profile_report().items():
            kern[k_] = (kern.get(k_, (0, 0.0))[0] + c_, kern.get(k_, (0, 0.0))[1] + m_)
        e_.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        sec = dt / a.steps
        in_bytes = pipe.total * world
        # algorithmic bytes per launch: fragment/hash kernels read every input byte once;
        # the LZ77 and checksum kernels read every unique byte once (+ r bytes written)
        ub = pipe.stats["unique_bytes"]
        alg = {"fragment_spec_kernel": pipe.total, "sha1_extents_kernel": pipe.total, "lz77_spec_kernel": ub + out_bytes,
               "sha1_chain_kernel": ub, "fragment_stitch_kernel": None, "lz77_stitch_kernel": None}
        # Kernels that occupy a handful of waves (one wave per 16 MiB block / per 1 MiB LZ segment): latency-bound
        # serial chains that run beside the chip-wide kernels of the next step.  They are listed in roofline_all
        # (with their wave count) but the headline roofline is the chip-wide kernel that holds the GPU longest.
        few_waves = {"sha1_chain_kernel": pipe.stats["blocks"], "lz77_spec_kernel": -(-ub // (1 << 20)),
                     "lz77_seam_kernel": -(-ub // (1 << 20)), "lz77_stitch_kernel": pipe.stats["blocks"]}
        traffic = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")     # PMC bytes per launch from the last rocprofv3 --pmc run
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get("bytes_per_launch", {})

        def roof(k):
            cnt, ms = kern[k]
            per = ms / cnt
            ab = alg.get(k)
            if not ab:
                return None
            ach = ab / 1e9 / (per / 1e3)
            r = {"bound": "hbm", "kernel": k, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(k), "avg_launch_ms": round(per, 4),
                 "algorithmic_bytes_per_launch": int(ab)}
            if k in few_waves:
                r["waves"] = int(few_waves[k])
                r["note"] = "latency-bound serial chain on %d waves of 1024 SIMDs; overlaps the next step" % few_waves[k]
            return r
        dom = max((k for k in kern if alg.get(k) and k not in few_waves), key=lambda k: kern[k][1], default=None)
        roof_dom = roof(dom) if dom else None
        roof_all = [r for r in (roof(k) for k in sorted(kern, key=lambda k: -kern[k][1])) if r]
        res = {"metric": "MB/s compressed output (bit-identical .zpaq) at -m1, Silesia x256", "value": round(out_bytes / 1e6 / sec, 3),
               "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "silesia_x%d_m1" % a.copies, "files": pipe.nfiles * world, "input_bytes": in_bytes,
                          "method": "14 -> x4,1,5,0,3,24", "block_bytes": BLOCK_LIMIT, "fragment": 6, **pipe.stats},
               "input_GBps": round(in_bytes / 1e9 / sec, 3), "steps_in_flight": depth,
               "ms_per_step_serial": round(stagger[0] * 1e3, 3) if depth > 1 else round(sec * 1e3, 3),
               "kernels_ms_per_step": {k: round(v[1] / a.steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
               "roofline": roof_dom, "roofline_all": roof_all}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(corpus, a.copies)
        if not a.no_verify and getattr(pipe, "verify_sample", None) is not None:
            import orc
            bin_, bout, nm = pipe.verify_sample
            want, _ = orc.compress_block(bytes(bin_.cpu().numpy()), "14", nm.decode(), "jDC\x01", True)
            res["verified_block0_bit_identical"] = bool(want == bytes(bout.cpu().numpy()))
        print(json.dumps(res))
    if a.dump_archive and rank == 0:
        parts = pipe.gathered if world > 1 else [pipe.verify_sample_all]
        # blocks are owned in ascending order by ascending rank: concatenating the per-rank streams IS block order
        with open(a.dump_archive, "wb") as f:
            for g in parts:
                f.write(bytes(g.cpu().numpy()))
    if world > 1:
        dist.destroy_process_group()
    for e_ in engines:
        e_.close()


if __name__ == "__main__":
    main()
