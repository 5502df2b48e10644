#!/usr/bin/env python3
"""bench.py -- zpaqfranz's block compress / decompress hot path on MI355X, whole job per step, inputs resident in HBM.

Workloads (BASELINE.json `configs`):
  silesia_x256_m1 (default, configs[1])  add -m1 of the Silesia corpus replicated x256 (3072 files, 54 256 276 480 bytes):
                  fragment -> SHA-1 -> dedup -> pack -> LZ77 level 1 -> ZPAQ block framing.  Method "14" -> x4,1,5,0,3,24.
  dup8_m1         (configs[3] at the largest single-GPU size) 1024 unique 16 MiB units, every unit stored 8 times in
                  shuffled order (128 GiB in 2048 files of 64 MiB): the compressor is the workload (~1000 d blocks).
  text_m2         (configs[2]) add -m2 of 10^9 bytes of text: 15 blocks of 64 MiB, each through the suffix array and the
                  LZ77-SA parse (method 2 -> "x6,1,4,0,7,27,1": 127 suffix-array neighbours either side, one byte of
                  lookahead; no context model at this level).  enwik9 itself cannot be fetched: a seeded word-model text
                  generated on the device stands in.
  extract_m1      (configs[4]) extract of the silesia_x256_m1 archive: every d block decoded (device-resident
                  Decompresser), every fragment's SHA-1 checked against the h table, the 3072 files assembled in HBM
                  and their SHA-256 compared with the originals'.
The real corpus cannot be fetched here: a seeded synthetic corpus with Silesia's member names and sizes stands in
("data": "synthetic"; its -m1 ratio is 0.31, and MB/s of OUTPUT scales with that ratio).

One JSON line is printed by rank 0 (contract in the task statement).  metric/unit are BASELINE.json's: MB/s of
compressed archive output (for extract: of compressed archive input); the input-side GB/s is reported next to it
because dedup collapses the x256 corpus to one copy before compression.

Every workload verifies ALL of its results against the CPU oracle outside the timed region: fragment boundaries
and SHA-1s, the dedup map, every d block byte for byte (add); every file's SHA-256 against hashlib over the
originals (extract).

N > 1 (torchrun, one rank per GPU, RCCL): weak scaling -- every rank owns its own corpus (different seed), fragments
and hashes it, the fragment tables are all-gathered over RCCL, every rank resolves the global first-occurrence dedup
identically, blocks are packed by the one deterministic global packer and owned by the rank that holds their first
fragment (seam fragments travel peer to peer), and the compressed blocks are all-gathered so that rank 0 could stitch
the archive in fixed block order.  `--force-collectives` runs that code path over RCCL with a single rank."""
import argparse
import ctypes as C
import json
import os
import sys
import time

# Every step in flight owns three HIP streams (engine main + checksum chains, torch); the runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue serialise -- a 216 ms checksum chain
# then holds up another step's kernels.  One queue per stream (must be set before the HIP runtime starts): 32 for the twelve
# steps in flight of round 4 (measured twice each: 87.6 / 89.5 ms per step with 32 queues, 95.8 / 98.3 with 16).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from zpaqfranz_amd.sharding import BLOCK_LIMIT, plan as shard_plan
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
# integer-issue ceiling (the real bound of the hash/fragment passes).  Measured with the SQ counters
# (profiles/r02_pmc_sq.json): SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4 cycles per wave64 integer instruction on every
# kernel of this path, one wave or many per SIMD -- 16 lanes per clock per SIMD: 256 CUs x 4 SIMDs x 16 x 2.4 GHz
LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9
# wave-instructions per input byte x 64 lanes (PMC: SQ_INSTS_VALU / bytes): lane-per-extent kernels use every lane,
# the wave-per-chain kernels one lane's worth of work per instruction
VALU_OPS_PER_BYTE = {"sha1_extents_kernel": 10.5, "fragment_spec_kernel": 13.7, "sha256_chain_kernel": 14.8 * 64, "sha1_chain_kernel": 6.35 * 64,
                     "sha256_extents_kernel": 21.5, "blake3_chunks_kernel": 12.0}


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
    the software-pipeline order  (k,0) (k,1) (k-lag, 2)  for k = 0, 1, ...: a fixed function of (steps, depth),
    hence identical on every rank, and one that lets step k exchange its tables before step k-1 has finished
    compressing.  A section holds the turn until its transfers have completed.
    lag: how many later steps' tables go out before a step's streams.  With F = a step's time to its tables, C = its compress time
    and P the period between steps, the order costs  P >= C / (lag + 1)  (step k+1's tables wait for step k-lag's streams) and
    P >= F / (depth - lag)  (a worker is free again only after step k-lag's streams, which wait for step k's tables): lag =
    depth - 1, the choice until round 6, makes the second bound P >= F -- one step per fragment pass, 297 ms measured through the
    product call (profiles/r06c) -- depth // 2 balances the two (F ~ C under load)."""

    def __init__(self, steps, depth, sections=3, lag=None):
        import threading
        self.seq = []
        lag = max(1, depth // 2) if lag is None else lag
        lag = min(lag, max(0, depth - 1))
        # (sections > 3: the product's zpqj_add_sharded_dev has two late sections -- the sizes of the compressed blocks as a host
        #  string, then the blocks themselves through the device form of the collective)
        for k in range(steps + depth):
            for st, sec in [(k, 0), (k, 1)] + [(k - lag, q) for q in range(2, sections)]:
                if 0 <= st < steps:
                    self.seq.append((st, sec))
        self.head = 0
        self.failed = False
        self.cv = threading.Condition()

    def abort(self):                   # a step failed on some thread: nobody waits for its turn any more
        with self.cv:
            self.failed = True
            self.cv.notify_all()

    def enter(self, step, sec, timeout=600.0):
        with self.cv:
            t0 = time.monotonic()
            while self.seq[self.head] != (step, sec):
                if self.failed:
                    raise RuntimeError("another step in flight failed")
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


# ---------------------------------------------------------------------------------------------------------------------
# corpora (synthetic stand-ins, resident in HBM)
# ---------------------------------------------------------------------------------------------------------------------
def steady_window(run_steps, barrier, depth, warm, steps, before=lambda: None):
    """The timed region.  With ONE step in flight: barrier, K steps, barrier (the contract, literally).  With `depth` steps in flight
    (software pipeline over `depth` contexts) a barrier on both sides of K steps also times filling and draining the pipeline, which
    makes the figure a function of K (162 ms at --steps 20, 100 ms at 48, same tree, same chip).  So the
    pipeline is kept primed: ONE continuous run of  depth (prime) + W (warm-up) + K (timed) + depth (tail, keeps the chip as full at
    the end of the window as at its start)  jobs between two barriers; the clock starts when the (depth + W)-th job has COMPLETED
    (a job's completion is a host-side fact: the product call has returned with the archive in host memory) and stops when K further
    jobs have completed.  Exactly K jobs complete inside the window; nothing is skipped, every job is the full step.
    Returns (seconds for the K steps, jobs run between the barriers, result of the last job, completion-time list)."""
    if depth <= 1:
        run_steps(warm)
        before()
        barrier(); t0 = time.perf_counter()
        out, _ = run_steps(steps)
        barrier()
        return time.perf_counter() - t0, steps, out, None
    n = depth + warm + steps + depth
    before()
    barrier(); t0 = time.perf_counter()
    out, done = run_steps(n)
    barrier()
    ct = sorted(done)
    k0 = depth + warm
    return ct[k0 + steps - 1] - ct[k0 - 1], n, out, [round(t - t0, 4) for t in ct]


STEADY_NOTE = ("steady state: one continuous run of depth + warmup + steps + depth jobs between two barriers; the clock runs from the completion of "
               "job depth+warmup to the completion of job depth+warmup+steps (exactly `steps` whole jobs complete inside it, the pipeline full at both "
               "ends); ms_per_step_cold = barrier, `steps` jobs from an idle chip, barrier (includes filling and draining the pipeline)")


def silesia_layout(dev, corpus, copies):
    """x`copies` replication of the 12-member corpus: files in copy order, one flat device buffer."""
    base = b"".join(b for _, b in corpus)
    sizes = [len(b) for _, b in corpus]
    unit = len(base)
    total = unit * copies
    data = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    data[:unit].copy_(torch.frombuffer(bytearray(base), dtype=torch.uint8))
    for c in range(1, copies):
        data[c * unit:(c + 1) * unit].copy_(data[:unit])
    data[total:].zero_()
    off = [0]
    for c in range(copies):
        for s in sizes:
            off.append(off[-1] + s)
    return dict(data=data, file_off=off, total=total, unit=unit, sizes=sizes, copies=copies, kind="silesia", names=[n for n, _ in corpus])


UNIT = 1 << 24


def dup8_units(pool_np, k):
    """unique unit k of the dup8 workload: pool unit k % npool with every byte rotated by k // npool (a bijection on byte
    values: same redundancy, different content, hence different fragment boundaries and ids)."""
    npool = len(pool_np) >> 24
    return (pool_np[(k % npool) << 24:((k % npool) + 1) << 24] + np.uint8(k // npool)).astype(np.uint8)


def dup8_layout(dev, corpus, units, dup, seed):
    """`units` unique 16 MiB units (SURVEY.md 8d-4 shape: LZ-compressible units, here byte-rotated stretches of the
    Silesia-shaped corpus), each stored `dup` times in shuffled order (seed 0xD00D + rank), files of 4 units."""
    base = b"".join(b for _, b in corpus)
    npool = len(base) >> 24
    pool = torch.frombuffer(bytearray(base[: npool << 24]), dtype=torch.uint8).to(dev).view(npool, UNIT)
    nslots = units * dup
    total = nslots * UNIT
    data = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    order = np.random.default_rng(0xD00D + seed).permutation(nslots) % units
    for s, u in enumerate(order.tolist()):
        torch.add(pool[u % npool], u // npool, out=data[s * UNIT:(s + 1) * UNIT])      # uint8 arithmetic wraps
    data[total:].zero_()
    per_file = 4
    off = [min(i * per_file * UNIT, total) for i in range((nslots + per_file - 1) // per_file + 1)]
    return dict(data=data, file_off=off, total=total, kind="dup8", units=units, dup=dup, order=order, npool=npool,
                pool_bytes=base[: npool << 24])


# ---------------------------------------------------------------------------------------------------------------------
# add pipeline
# ---------------------------------------------------------------------------------------------------------------------
class Pipeline:
    def __init__(self, eng, device, layout, rank, world, collectives=False):
        from zpaqfranz_amd import engine as E
        self.E, self.eng, self.dev, self.rank, self.world = E, eng, device, rank, world
        self.coll = collectives or world > 1
        self.layout = layout
        self.data = layout["data"]                       # the input is read-only: pipelines share it
        self.total = layout["total"]
        self.file_off = layout["file_off"]
        self.nfiles = len(self.file_off) - 1
        self.params = eng.fragment_params()
        self.cap = eng.fragment_capacity(self.file_off, self.params)
        i64, i32, u8 = torch.int64, torch.int32, torch.uint8
        self.frag_off = torch.empty(self.cap, dtype=i64, device=device)
        self.frag_len = torch.empty(self.cap, dtype=i32, device=device)
        self.frag_file = torch.empty(self.cap, dtype=i32, device=device)
        self.digests = torch.empty(self.cap * 20 + 64, dtype=u8, device=device)
        self.first = torch.empty(self.cap * max(1, world), dtype=i32, device=device)
        self.tstream = torch.cuda.Stream(device=device)
        self.keep_outputs = True
        self.use_twins = True        # zpq_fragment_sha1_dev with the twin-file fold (False: the two plain calls, every byte hashed)
        self.twin_stats = None
        torch.cuda.synchronize()

    VERSION_DATE = 20240101000000

    def file_names(self):
        """names in the order the files lie in HBM, ascending (what zpqj_add_dev asks for)"""
        L = self.layout
        if L["kind"] == "silesia":
            return ["c%04d/%s" % (c, n) for c in range(L["copies"]) for n in L["names"]]
        return ["f%06d" % i for i in range(self.nfiles)]

    def step_product(self, keep=True):
        """One C-ABI call: zpqj_add_dev (shim/jidac_gpu.cpp) -- device-resident files in, the whole journaling archive (c, d, h, i
        blocks) out; the product's own orchestration of what step() does call by call.  Returns the archive's length."""
        E = self.E
        if getattr(self, "devfiles", None) is None:
            self.devfiles = E.DevFiles(self.file_names(), self.file_off, version_date=self.VERSION_DATE)
        ptr, n, st = E.jidac_add_dev(self.eng, b"", self.data.data_ptr(), self.devfiles, self.VERSION_DATE, "14", twins=self.use_twins, raw=True)
        self.drop_archive()
        if keep:       # no copy (config 4's archive is several GB): a view of the memory the call returned, released by drop_archive()
            self.archive_ptr = ptr
            self.archive = memoryview((C.c_ubyte * n).from_address(ptr)).cast("B")
        else:
            E.load_shim().zpqj_free(C.c_void_p(ptr))
        self.stats = dict(fragments=st["fragments"], unique_fragments=st["new_fragments"], blocks=st["d_blocks"], unique_bytes=st["unique_bytes"],
                          out_bytes=int(n), d_bytes=st["d_bytes"])
        return int(n)

    def drop_archive(self):
        if getattr(self, "sharded", False):
            self.archive = None
            return
        if getattr(self, "archive_ptr", None):
            self.archive = None
            self.E.load_shim().zpqj_free(C.c_void_p(self.archive_ptr))
            self.archive_ptr = None

    def use_sharded_product(self, names, sizes, gather, gather_dev):
        """Several ranks: the timed step becomes ONE C-ABI call per rank, zpqj_add_sharded_dev -- this rank's files resident in HBM in,
        the whole job's archive out on every rank -- with the in-tree RCCL collectives (shim/rccl_gather.cpp) inside it: no torch
        collective in the timed region.  names / sizes: every file of the job, the same lists on every rank."""
        self.all_names, self.all_sizes, self.gather, self.gather_dev = names, sizes, gather, gather_dev
        self.sharded = True

    SECTIONS = 4          # collectives of one zpqj_add_sharded_dev: fragment tables, seam fragments, block sizes, blocks (device form)

    def step_sharded(self, order, idx, keep=True):
        E = self.E

        def wrap(k, thunk):           # every collective of the call takes its turn in the one order all ranks share
            order.enter(idx, k)
            try:
                return thunk()
            finally:
                order.leave()
        arc, st = E.jidac_add_sharded_dev(self.eng, self.rank, self.world, self.gather, self.gather_dev, None, self.all_names, self.all_sizes,
                                          self.data.data_ptr(), self.VERSION_DATE, "14", twins=self.use_twins,
                                          wrap=None if isinstance(order, _NoOrder) else wrap)
        self.archive = arc if keep else None
        self.stats = dict(fragments=st["fragments"], unique_fragments=st["new_fragments"], blocks=st["d_blocks"], unique_bytes=st["unique_bytes"],
                          out_bytes=len(arc), d_bytes=st["d_bytes"])
        return len(arc)

    def step(self, order=None, idx=0, keep=True):
        """keep: hold on to the block inputs / outputs of this step for the verification (costs their memory until the next step)"""
        if getattr(self, "sharded", False):
            return self.step_sharded(order or _NoOrder(), idx, keep)
        if getattr(self, "product", False):
            return self.step_product(keep)
        self.keep_outputs = keep
        if not keep:
            self.verify_blocks = None
        return self.phase_b(self.phase_a(), order or _NoOrder(), idx)

    def phase_a(self):
        """Local half of a step: fragment + SHA-1 of every fragment.  No collective, so a helper thread may run it
        for step i+1 while the main thread is in phase_b of step i (the multi-rank pipeline)."""
        eng = self.eng
        with torch.cuda.stream(self.tstream):
            if self.use_twins:
                # one call: files equal to an earlier file are found by comparison, the rest is fragmented and hashed
                nf = eng.fragment_sha1_dev(self.data.data_ptr(), self.file_off, self.params, self.frag_off.data_ptr(),
                                           self.frag_len.data_ptr(), self.frag_file.data_ptr(), self.digests.data_ptr(), self.cap)
                self.twin_stats = eng.last_twin_stats
            else:
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
        if self.coll:
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
        P = shard_plan(first, lens, cnts, self.rank, balance=getattr(self, 'balance_blocks', False))
        uniq_idx, mine, starts, nblk = P["uniq_idx"], P["mine"], P["starts"], P["nblocks"]
        # block buffers of the blocks this rank owns: fragments + size table + 0 + count
        blk_n, src_off, src_len, dst_off, trailers, layout = [], [], [], [], [], {}
        pos = 0
        for b in mine:
            u, l, own_rank, src = P["blocks"][int(b)]      # src: the occurrence each fragment is read from (this rank's own copy where it has one)
            o = pos + np.concatenate(([0], np.cumsum(l)[:-1]))
            own = own_rank == self.rank
            src_off.append(src[own] - my_lo); src_len.append(l[own]); dst_off.append(o[own])
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
            if so.numel():        # (a block dealt to this rank may hold none of its own fragments: --shared-corpus)
                eng.gather_dev(self.data.data_ptr(), abs_off.data_ptr(), sl.data_ptr(), do.data_ptr(), so.numel(),
                               blocks_buf.data_ptr())
            if len(trailers) <= 64:
                for p_, tr in trailers:
                    blocks_buf[p_:p_ + len(tr)] = torch.frombuffer(bytearray(tr), dtype=torch.uint8).to(dev)
            else:
                # size tables of all blocks: one host buffer, one upload, one scatter
                tr_all = np.frombuffer(b"".join(tr for _, tr in trailers), dtype=np.uint8)
                tr_pos = np.concatenate([np.arange(p_, p_ + len(tr), dtype=np.int64) for p_, tr in trailers])
                blocks_buf[torch.from_numpy(tr_pos).to(dev)] = torch.from_numpy(tr_all.copy()).to(dev)
        if self.coll:
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
            olen = [jobs[k].out_len for k in range(nb)]
            out_bytes = sum(olen)
            self.last_blocks = [(int(mine[k]), olen[k]) for k in range(nb)]
            if self.keep_outputs:
                # kept for the verification outside the timed region: inputs and framed outputs of every owned block
                in_off = np.concatenate(([0], np.cumsum([(n + 63 + 64) & ~63 for n in blk_n])))[:-1]
                out_off = np.concatenate(([0], np.cumsum(ocap)))[:-1]
                self.verify_blocks = dict(buf=blocks_buf, n=list(blk_n), in_off=in_off.tolist(), outs=outs, out_off=out_off.tolist(),
                                          out_len=olen, names=names)
        if self.coll:
            # the archive is stitched on rank 0 in block order: gather the compressed streams
            order.enter(idx, 2)
            t = torch.tensor([out_bytes], dtype=torch.int64, device=dev)
            ts = _all_gather(t)
            mx = max(int(x.item()) for x in ts)
            buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
            if outs is not None:
                q = 0; p_out = 0
                for k in range(nb):
                    buf[q:q + olen[k]] = outs[p_out:p_out + olen[k]]
                    q += olen[k]; p_out += ocap[k]
            gathered = _all_gather(buf)
            out_bytes = sum(int(x.item()) for x in ts)
            self.gathered = [g[: int(x.item())] for g, x in zip(gathered, ts)]   # rank r's framed blocks, block order
            tsync()
            order.leave()
        self.last = dict(nf=nf, ntot=ntot, first=first, lens=lens, plan=P, my_lo=my_lo)
        self.stats = dict(fragments=int(ntot), unique_fragments=int(len(uniq_idx)), blocks=int(nblk),
                          unique_bytes=int(lens[uniq_idx].sum()), out_bytes=int(out_bytes))
        return out_bytes

    def framed_blocks(self):
        """[(first fragment id, framed d block bytes on the host)] of the last step, block order."""
        v = self.verify_blocks
        return [(v["names"][k], bytes(v["outs"][v["out_off"][k]:v["out_off"][k] + v["out_len"][k]].cpu().numpy())) for k in range(len(v["n"]))]

    def _exchange_seams(self, P, lens, my_lo, nf, layout, blocks_buf):
        """Fragments of a block that live on another rank (only at rank seams) travel peer to peer:
        both sides derive the same ordered lists from the plan, so one send/recv per rank pair suffices."""
        dev, eng = self.dev, self.eng
        i64, i32 = torch.int64, torch.int32
        sends, recvs = {}, {}
        for dst, idx in P["send"].items():
            # one gather kernel packs the outgoing fragments back to back (no per-fragment host work)
            loc = torch.from_numpy((idx - my_lo).astype(np.int64)).to(dev)
            ln = torch.from_numpy(lens[idx].astype(np.int32)).to(dev)
            src = self.frag_off[:nf][loc].contiguous()
            dst_off = (torch.cumsum(ln.to(i64), 0) - ln.to(i64)).contiguous()
            buf = torch.empty(int(lens[idx].sum()) + 64, dtype=torch.uint8, device=dev)
            torch.cuda.current_stream().synchronize()
            eng.gather_dev(self.data.data_ptr(), src.data_ptr(), ln.data_ptr(), dst_off.data_ptr(), len(idx), buf.data_ptr())
            eng.sync()
            sends[int(dst)] = buf[: int(lens[idx].sum())]
        for src, idx in P["recv"].items():
            recvs[int(src)] = int(lens[idx].sum())
        got = _exchange(sends, recvs, dev)
        for src, idx in P["recv"].items():
            buf = got[int(src)]
            ln = torch.from_numpy(lens[idx].astype(np.int32)).to(dev)
            src_off = (torch.cumsum(ln.to(i64), 0) - ln.to(i64)).contiguous()
            dst_off = torch.tensor([int(layout[int(g)]) for g in idx.tolist()], dtype=i64, device=dev)
            pad = torch.cat([buf, torch.zeros(64, dtype=torch.uint8, device=dev)])      # the library reads up to 64 bytes past an extent
            torch.cuda.current_stream().synchronize()
            eng.gather_dev(pad.data_ptr(), src_off.data_ptr(), ln.data_ptr(), dst_off.data_ptr(), len(idx), blocks_buf.data_ptr())
            eng.sync()


def verify_add(pipe, layout, corpus, threads):
    """Everything the last step produced, against the CPU oracle (outside the timed region): fragment boundaries and ids
    of every file, the first-occurrence map, every d block byte for byte.  Single-rank runs only."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    import orc
    L = pipe.last
    nf = L["nf"]
    foff = pipe.frag_off[:nf].cpu().numpy(); flen = pipe.frag_len[:nf].cpu().numpy().astype(np.int64); ffile = pipe.frag_file[:nf].cpu().numpy()
    dig = pipe.digests[: nf * 20].cpu().numpy().reshape(nf, 20)
    file_off = np.array(layout["file_off"], dtype=np.int64)
    res = {}
    # distinct file contents: the 12 members (silesia) or every file (dup8, sampled by the oracle below)
    if layout["kind"] == "silesia":
        members = [b for _, b in corpus]

        def member_digests(b, w):
            offs = np.concatenate(([0], np.cumsum(w)))[:-1].tolist()
            return [hashlib.sha1(b[o:o + l]).digest() for o, l in zip(offs, w)]
        with ThreadPoolExecutor(threads) as ex:
            want = list(ex.map(orc.chunk, members))
            want_dig = list(ex.map(member_digests, members, want))
        nm = len(members)
        ok_len = ok_dig = True
        bounds = np.searchsorted(ffile, np.arange(pipe.nfiles + 1))
        want_arr = [np.array(w, dtype=np.int64) for w in want]
        want_da = [np.frombuffer(b"".join(d), dtype=np.uint8).reshape(-1, 20) for d in want_dig]
        for f in range(pipe.nfiles):
            a, b = bounds[f], bounds[f + 1]
            m = f % nm
            if b - a != len(want_arr[m]) or not np.array_equal(flen[a:b], want_arr[m]) or foff[a] != file_off[f]:
                ok_len = False; break
            if not np.array_equal(dig[a:b], want_da[m]):
                ok_dig = False; break
        res["verified_fragments"] = bool(ok_len and ok_dig)
    else:
        # dup8: every file is 4 units; the oracle fragments and hashes a sample of files (all of them would be 128 GiB of CPU work; blocks and the dedup map below ARE checked completely)
        rng = np.random.default_rng(1)
        sample = sorted(set(rng.integers(0, pipe.nfiles, 256).tolist()))    # ~11 % of the files (17 GB through the oracle)
        pool = np.frombuffer(layout["pool_bytes"], dtype=np.uint8)
        bounds = np.searchsorted(ffile, np.arange(pipe.nfiles + 1))

        def one(f):
            slots = range(f * 4, min(f * 4 + 4, len(layout["order"])))
            content = b"".join(dup8_units(pool, int(layout["order"][s])).tobytes() for s in slots)
            w = orc.chunk(content)
            offs = np.concatenate(([0], np.cumsum(w)))[:-1].tolist()
            d = np.frombuffer(b"".join(hashlib.sha1(content[o:o + l]).digest() for o, l in zip(offs, w)), dtype=np.uint8).reshape(-1, 20)
            a, b = bounds[f], bounds[f + 1]
            return b - a == len(w) and np.array_equal(flen[a:b], np.array(w)) and np.array_equal(dig[a:b], d)
        with ThreadPoolExecutor(threads) as ex:
            res["verified_fragments"] = bool(all(ex.map(one, sample)))
        res["verified_fragments_sample"] = "%d of %d files" % (len(sample), pipe.nfiles)
    # dedup: exact first-occurrence map over the digests
    seen = {}
    first_want = np.empty(nf, dtype=np.int64)
    keys = dig.view([("k", "V20")]).ravel()
    _, idx, inv = np.unique(keys, return_index=True, return_inverse=True)
    first_want = idx[inv]
    res["verified_dedup"] = bool(np.array_equal(first_want, L["first"][:nf].astype(np.int64)))
    # every d block, byte for byte
    v = pipe.verify_blocks
    nb = len(v["n"])

    def blk(k):
        bin_ = bytes(v["buf"][v["in_off"][k]:v["in_off"][k] + v["n"][k]].cpu().numpy())
        bout = bytes(v["outs"][v["out_off"][k]:v["out_off"][k] + v["out_len"][k]].cpu().numpy())
        want, _ = orc.compress_block(bin_, "14", v["names"][k].decode(), "jDC\x01", True)
        return want == bout
    with ThreadPoolExecutor(threads) as ex:
        oks = list(ex.map(blk, range(nb)))
    res["verified_all_blocks"] = bool(all(oks))
    res["verified_blocks"] = "%d of %d" % (sum(oks), nb)
    return res


def split_archive(arc, payloads=None):
    """payloads: block types ("chi") whose stored bytes are wanted (default: all).  [(name, comment, start, end, payload bytes)] of the blocks of a journaling archive whose blocks have no context model
    (stored sub-blocks): tag, zPQ level type, header, 1 name 0 comment 0 0, {len[4] bytes}... 0[4], 253 sha1[20] | 254, 255."""
    arc = memoryview(arc).cast("B")              # (bytes or the view of a multi-gigabyte archive: nothing below copies more than a block)
    out, p, n = [], 0, len(arc)

    def nul(q):
        return q + bytes(arc[q:q + 4096]).index(0)
    while p < n:
        s0 = p
        assert bytes(arc[p + 13:p + 16]) == b"zPQ", "no block at %d" % p
        hs = arc[p + 18] | arc[p + 19] << 8
        assert arc[p + 24] == 0, "block with a context model"
        p += 20 + hs
        assert arc[p] == 1
        e = nul(p + 1); name = bytes(arc[p + 1:e])
        e2 = nul(e + 1); comment = bytes(arc[e + 1:e2])
        p = e2 + 2
        pay = bytearray() if payloads is None or chr(name[17]) in payloads else None
        while True:
            k = int.from_bytes(arc[p:p + 4], "big"); p += 4
            if not k:
                break
            if pay is not None:
                pay += arc[p:p + k]
            p += k
        p += 21 if arc[p] == 253 else 1
        assert arc[p] == 255
        p += 1
        out.append((name, comment, s0, p, bytes(pay) if pay is not None else None))
    return out


def verify_product(pipe, archive, threads):
    """The archive zpqj_add_dev returned in the timed region against (a) the call-by-call orchestration of the same job (pipe.last /
    pipe.verify_blocks, which verify_add checks against the oracle): its d blocks byte for byte, its h blocks = the fragment
    ids and sizes, its i blocks = every file's name and pointer list; (b) the REAL reference decoder, which walks every c, h
    and i block (the d blocks are the bytes of (a))."""
    import orc
    res = {}
    blocks = split_archive(archive, payloads="chi")
    kinds = "".join(chr(b[0][17]) for b in blocks)
    nb = kinds.count("d")
    res["archive_blocks"] = {k: kinds.count(k) for k in "cdhi"}
    ok_layout = kinds == "c" + "d" * nb + "h" * nb + "i" * kinds.count("i") and kinds.count("i") >= 1
    want = pipe.framed_blocks()
    got_d = [archive[b[2]:b[3]] for b in blocks if chr(b[0][17]) == "d"]        # (views)
    res["verified_product_d_blocks"] = bool(ok_layout and len(want) == len(got_d) and all(w[0] == b[0] and w[1] == g for w, g, b in
                                                                                       zip(want, got_d, [b for b in blocks if chr(b[0][17]) == "d"])))
    # reference decoder over the index blocks
    ok_ref = True
    if orc.have_ref():
        for b in blocks:
            if chr(b[0][17]) == "d":
                continue
            r = orc.ref_decompress_block(bytes(archive[b[2]:b[3]]), len(b[4]) * 8 + 65536)
            ok_ref = ok_ref and r["sha1_ok"] == 1 and r["consumed"] == b[3] - b[2] and r["filename"] == b[0]
    res["verified_index_blocks_by_reference_decoder"] = bool(ok_ref and orc.have_ref())
    L = pipe.last
    nf, first, lens, P = L["nf"], L["first"][:L["nf"]].astype(np.int64), L["lens"], L["plan"]
    uniq = P["uniq_idx"]
    fid = np.zeros(nf, dtype=np.int64); fid[uniq] = 1 + np.arange(len(uniq)); fid = fid[first]      # fragment id of every file fragment
    dig = pipe.digests[: nf * 20].cpu().numpy().reshape(nf, 20)
    # c block: the d blocks' bytes; h blocks: bsize + (sha1, usize) per fragment of its d block
    cb = [b for b in blocks if chr(b[0][17]) == "c"][0]
    ok = ok_layout and int.from_bytes(orc.ref_decompress_block(bytes(archive[cb[2]:cb[3]]), 65536)["data"] if orc.have_ref() else b"", "little") == sum(len(g) for g in got_d)
    st = P["starts"]
    hb = [b for b in blocks if chr(b[0][17]) == "h"]
    for k, b in enumerate(hb):
        body = orc.ref_decompress_block(bytes(archive[b[2]:b[3]]), len(b[4]) * 2 + 65536)["data"] if orc.have_ref() else b""
        u = uniq[st[k]:st[k + 1]]
        wantb = len(got_d[k]).to_bytes(4, "little") + b"".join(bytes(dig[i]) + int(lens[i]).to_bytes(4, "little") for i in u.tolist())
        ok = ok and body == wantb and int(b[0][18:]) == int(st[k]) + 1
    res["verified_product_h_blocks"] = bool(ok)
    # i blocks: date[8] name 0 na[4] attr ni[4] ptr[ni][4] per file, in name order
    names = pipe.file_names()
    ffile = pipe.frag_file[:nf].cpu().numpy()
    bounds = np.searchsorted(ffile, np.arange(pipe.nfiles + 1))
    ok, f = True, 0
    for b in blocks:
        if chr(b[0][17]) != "i":
            continue
        body = orc.ref_decompress_block(bytes(archive[b[2]:b[3]]), 1 << 22)["data"] if orc.have_ref() else b""
        q = 0
        while q < len(body) and ok:
            e = body.index(b"\0", q + 8)
            na = int.from_bytes(body[e + 1:e + 5], "little")
            ni = int.from_bytes(body[e + 5 + na:e + 9 + na], "little")
            ptr = np.frombuffer(body, dtype="<u4", count=ni, offset=e + 9 + na)
            ok = f < pipe.nfiles and body[q + 8:e].decode() == names[f] and np.array_equal(ptr.astype(np.int64), fid[bounds[f]:bounds[f + 1]])
            q = e + 9 + na + 4 * ni
            f += 1
    res["verified_product_i_blocks"] = bool(ok and f == pipe.nfiles)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# extract pipeline (configs[4])
# ---------------------------------------------------------------------------------------------------------------------
class ExtractPipeline:
    """Jidac::extract over an archive staged in HBM (ZSFX/zsfx.cpp:1731-1994): every d block is decoded
    (zpq_decompress_blocks_dev, stored SHA-1 checked on the device), every fragment's SHA-1 is compared with the h
    table, the fragments are copied to their places in the output files, and every file's SHA-256 is compared with
    the original's.  The index (h / i blocks: fragment sizes and ids, file pointer lists) is host data, as in Jidac."""

    def __init__(self, eng, dev, add_pipe, layout, file_sha256):
        from zpaqfranz_amd import engine as E
        self.E, self.eng, self.dev = E, eng, dev
        L = add_pipe.last
        nf, first, lens, P = L["nf"], L["first"], L["lens"], L["plan"]
        v = add_pipe.verify_blocks
        nb = len(v["n"])
        # the archive's d blocks, back to back in HBM (64 bytes of padding after each)
        self.blk_len = list(v["out_len"])
        offs, pos = [], 0
        for n in self.blk_len:
            offs.append(pos); pos += (n + 64 + 63) & ~63
        self.arc = torch.zeros(pos + 64, dtype=torch.uint8, device=dev)
        for k in range(nb):
            self.arc[offs[k]:offs[k] + self.blk_len[k]] = v["outs"][v["out_off"][k]:v["out_off"][k] + v["out_len"][k]]
        self.arc_off = offs
        self.arc_bytes = sum(self.blk_len)
        self.usize = list(v["n"])                          # decoded size of every d block (comment "<usize> jDC\x01")
        uoffs, pos = [], 0
        for n in self.usize:
            uoffs.append(pos); pos += (n + 64 + 63) & ~63
        self.plain = torch.empty(pos + 64, dtype=torch.uint8, device=dev)
        self.plain_off = uoffs
        # index: unique fragment u -> (block, offset in block); file fragment i -> unique fragment first[i]
        uniq_idx, blk = P["uniq_idx"], P["blk"]
        ulen = lens[uniq_idx]
        within = np.zeros(len(uniq_idx), dtype=np.int64)
        st = P["starts"]
        for b in range(nb):
            within[st[b]:st[b + 1]] = np.concatenate(([0], np.cumsum(ulen[st[b]:st[b + 1]])))[:-1]
        upos = np.array(uoffs, dtype=np.int64)[blk] + within              # offset of every unique fragment in self.plain
        rank_of = np.full(nf, -1, dtype=np.int64); rank_of[uniq_idx] = np.arange(len(uniq_idx))
        src = upos[rank_of[first[:nf]]]                                    # source of every file fragment
        dst = add_pipe.frag_off[:nf].cpu().numpy().astype(np.int64)        # its place in the output (files back to back)
        i64, i32 = torch.int64, torch.int32
        self.n_ext = nf
        self.d_src = torch.from_numpy(src).to(dev)
        self.d_dst = torch.from_numpy(dst).to(dev)
        self.d_len = torch.from_numpy(lens[:nf].astype(np.int32)).to(dev)
        # h table: expected SHA-1 of every unique fragment, and where it sits in the decoded blocks
        self.nu = len(uniq_idx)
        self.d_upos = torch.from_numpy(upos).to(dev)
        self.d_ulen = torch.from_numpy(ulen.astype(np.int32)).to(dev)
        self.d_want = add_pipe.digests.view(-1)[: nf * 20].view(nf, 20)[torch.from_numpy(uniq_idx).to(dev)].contiguous().view(-1)
        self.d_want = torch.cat([self.d_want, torch.zeros(64, dtype=torch.uint8, device=dev)])
        self.d_got = torch.empty(self.nu * 20 + 64, dtype=torch.uint8, device=dev)
        # output files and their expected SHA-256
        self.file_off = layout["file_off"]
        self.nfiles = len(self.file_off) - 1
        self.total = layout["total"]
        self.out = torch.empty(self.total + 64, dtype=torch.uint8, device=dev)
        fo = np.array(self.file_off, dtype=np.int64)
        self.d_foff = torch.from_numpy(fo[:-1].copy()).to(dev)
        self.d_flen = torch.from_numpy(np.diff(fo)).to(dev)
        self.d_sha_want = torch.from_numpy(np.frombuffer(b"".join(file_sha256), dtype=np.uint8).copy()).to(dev)
        self.d_sha_got = torch.empty(self.nfiles * 32 + 64, dtype=torch.uint8, device=dev)
        self.jobs = (E.UnblockJob * nb)()
        for k in range(nb):
            self.jobs[k].in_ = self.arc.data_ptr() + offs[k]; self.jobs[k].n = self.blk_len[k]
            self.jobs[k].out = self.plain.data_ptr() + uoffs[k]; self.jobs[k].out_cap = self.usize[k] + 64
        self.nb = nb
        self.stats = dict(blocks=nb, fragments=int(nf), unique_fragments=int(self.nu), files=self.nfiles,
                          archive_bytes=int(self.arc_bytes), restored_bytes=int(self.total))
        torch.cuda.synchronize()

    def clone_for(self, eng):
        """A second extract job in flight on another engine context: shares the archive and the index (read-only), owns
        its decoded blocks, its restored files and its digests."""
        import copy
        c = copy.copy(self)
        c.eng = eng
        dev = self.dev
        c.plain = torch.empty_like(self.plain)
        c.out = torch.empty(self.total + 64, dtype=torch.uint8, device=dev)
        c.d_got = torch.empty_like(self.d_got)
        c.d_sha_got = torch.empty_like(self.d_sha_got)
        c.jobs = (self.E.UnblockJob * self.nb)()
        for k in range(self.nb):
            c.jobs[k].in_ = self.arc.data_ptr() + self.arc_off[k]; c.jobs[k].n = self.blk_len[k]
            c.jobs[k].out = c.plain.data_ptr() + self.plain_off[k]; c.jobs[k].out_cap = self.usize[k] + 64
        torch.cuda.synchronize()
        return c

    def use_product(self, archive):
        """The timed step becomes ONE C-ABI call, zpqj_extract_dev: the whole journaling archive (c, d, h, i blocks as zpqj_add_dev
        returned them) resident in HBM in; restored files + their SHA-256 left in HBM.  The index is read by the call itself."""
        self.full = torch.zeros(len(archive) + 64, dtype=torch.uint8, device=self.dev)
        self.full[: len(archive)] = torch.frombuffer(bytearray(archive), dtype=torch.uint8).to(self.dev)
        self.full_len = len(archive)
        names, off, st = self.E.jidac_extract_dev(self.eng, self.full.data_ptr(), self.full_len)       # plan call: sizes only
        if off != list(self.file_off):
            raise RuntimeError("extract: the archive's index does not describe the files that were added")
        self.product = True

    def step_product(self):
        E = self.E
        _, _, st = E.jidac_extract_dev(self.eng, self.full.data_ptr(), self.full_len, self.out.data_ptr(), self.total + 64,
                                       self.d_sha_got.data_ptr(), self.nfiles, twins=getattr(self, "use_twins", False))
        if st["files"] != self.nfiles or st["bytes"] != self.total or st["d_blocks"] != self.nb:
            raise RuntimeError("extract: %r" % st)
        self.twin_stats = dict(twins=st["twins"], twin_bytes=st["twin_bytes"], compared=0, compared_bytes=st["compared_bytes"])
        mism, _ = self.eng.digest_compare_dev(self.d_sha_got.data_ptr(), self.d_sha_want.data_ptr(), self.nfiles, 32)
        self.sha256_mismatches = int(mism)
        return self.full_len

    def step(self, order=None, idx=0):
        eng = self.eng
        self.out.zero_() if getattr(self, "scrub", False) else None
        if getattr(self, "product", False) and getattr(self, "verify_hash", "sha256") == "sha256":
            return self.step_product()
        rc = eng.decompress_blocks_dev(self.jobs, self.nb, True)
        bad = [k for k in range(self.nb) if self.jobs[k].status != 0 or self.jobs[k].out_len != self.usize[k]]
        if rc != 0 or bad:
            raise RuntimeError("extract: d block decode failed (rc %d, blocks %r)" % (rc, bad[:5]))
        # fragment checksums against the h table (decompressThread, ZSFX/zsfx.cpp:1811-1834)
        eng.sha1_extents_dev(self.plain.data_ptr(), self.d_upos.data_ptr(), self.d_ulen.data_ptr(), self.nu, self.d_got.data_ptr())
        mism, _ = eng.digest_compare_dev(self.d_got.data_ptr(), self.d_want.data_ptr(), self.nu, 20)
        if mism:
            raise RuntimeError("extract: %d fragment checksums differ" % mism)
        # every file fragment to its place (the writes of ZSFX/zsfx.cpp:1880-1960, into HBM instead of the file system)
        eng.gather_dev(self.plain.data_ptr(), self.d_src.data_ptr(), self.d_len.data_ptr(), self.d_dst.data_ptr(), self.n_ext, self.out.data_ptr())
        if getattr(self, "verify_hash", "sha256") == "blake3":
            # the file-level hash zpaqfranz itself offers for this (README.md:95-105): a tree, so one long file is chip-wide work
            _, _, got = eng.file_checksums_dev(self.out.data_ptr(), self.file_off, crc32=False, xxh64=False, blake3=True)
            self.blake3_mismatches = sum(1 for g_, w_ in zip(got, self.blake3_want) if g_ != w_)
            return self.arc_bytes
        # SHA-256 of every restored file against the original's
        if getattr(self, "use_twins", True):
            # restored files whose bytes equal an earlier restored file's (compared on the device) take its digest
            self.twin_stats = eng.sha256_files_dev(self.out.data_ptr(), self.file_off, self.d_sha_got.data_ptr())
        else:
            eng.sha256_extents_dev(self.out.data_ptr(), self.d_foff.data_ptr(), self.d_flen.data_ptr(), self.nfiles, self.d_sha_got.data_ptr())
        mism, first = eng.digest_compare_dev(self.d_sha_got.data_ptr(), self.d_sha_want.data_ptr(), self.nfiles, 32)
        self.sha256_mismatches = int(mism)
        return self.arc_bytes


# ---------------------------------------------------------------------------------------------------------------------
def text_blocks_dev(dev, nbytes, seed, block=(1 << 26) - 4096):
    """Stand-in for enwik9: words drawn (Zipf) from a 2^18-word vocabulary over a skewed alphabet, with capitals,
    punctuation, line ends, numbers and markup-like tokens mixed in; generated on the device, one tensor per block
    (padded by 64 readable bytes)."""
    g = torch.Generator(device=dev); g.manual_seed(1234)            # the vocabulary is the same for every rank and block
    V = 1 << 18
    wl = (2 + torch.poisson(torch.full((V,), 4.2, device=dev), generator=g)).clamp_(1, 24).to(torch.int64)
    wl[:64] = torch.randint(1, 4, (64,), device=dev, generator=g)    # the most frequent words are short
    tail = torch.rand(V, device=dev, generator=g)
    sep = torch.full((V,), 32, dtype=torch.uint8, device=dev)
    sep[tail < 0.06] = 44; sep[tail < 0.03] = 46; sep[tail < 0.012] = 10          # ", " ". " newline
    vstart = torch.cumsum(wl + 1, 0) - (wl + 1)
    nlet = int((wl + 1).sum().item())
    freq = torch.tensor([8.2, 1.5, 2.8, 4.3, 12.7, 2.2, 2.0, 6.1, 7.0, .15, .77, 4.0, 2.4, 6.7, 7.5, 1.9, .1, 6.0, 6.3, 9.1, 2.8, .98, 2.4, .15, 2.0, .07], device=dev)
    flat = (97 + torch.multinomial(freq, nlet, replacement=True, generator=g)).to(torch.uint8)
    flat[vstart + wl] = sep
    cap = torch.rand(V, device=dev, generator=g) < 0.04
    flat[vstart[cap]] -= 32
    digits = torch.rand(V, device=dev, generator=g) < 0.02
    didx = torch.repeat_interleave(vstart[digits], wl[digits]) + (torch.arange(int(wl[digits].sum().item()), device=dev) -
                                                                  torch.repeat_interleave(torch.cumsum(wl[digits], 0) - wl[digits], wl[digits]))
    flat[didx] = (48 + torch.randint(0, 10, (didx.numel(),), device=dev, generator=g)).to(torch.uint8)
    w = 1.0 / (torch.arange(V, device=dev, dtype=torch.float64) + 2.7) ** 1.08
    cdf = torch.cumsum(w / w.sum(), 0)
    out = []
    g2 = torch.Generator(device=dev)
    for k in range((nbytes + block - 1) // block):
        n = min(block, nbytes - k * block)
        g2.manual_seed(seed * 1000 + k)
        nw = n // 5 + 64
        ids = torch.searchsorted(cdf, torch.rand(nw, device=dev, dtype=torch.float64, generator=g2)).clamp_(max=V - 1)
        ln = wl[ids] + 1
        ostart = torch.cumsum(ln, 0) - ln
        tot = int(ln.sum().item())
        assert tot >= n
        idx = torch.repeat_interleave(vstart[ids] - ostart, ln) + torch.arange(tot, device=dev)
        t = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        t[:n] = flat[idx[:n]]
        out.append((t, n))
        del idx, ids, ln, ostart
    return out


def main_text_m2(a, rank, world, local, dev):
    """configs[2]: every rank compresses its own 10^9 bytes of text with method 2's LZ77-SA (blocks are independent: no
    collective on the data path; weak scaling)."""
    from zpaqfranz_amd import Engine, engine as E
    method = b"2"                          # compressBlock: 64 MiB - 4096 byte blocks -> "x6,1,4,0,7,27,1"
    bs = (1 << 26) - 4096
    # steps in flight: a block's SHA-1 (64 MiB through one wave: 0.87 s) is as long as the whole LZ77 path of a step, and
    # both leave most of the chip idle at times -- three steps on three engine contexts overlap them (measured per
    # step: 777 ms with two, 732 with three, 739 with four)
    depth = max(1, a.pipeline if a.pipeline is not None else 3)
    engines = [Engine(local)]          # the others are created after one job has been timed alone (single_job)
    eng = engines[0]
    blocks = text_blocks_dev(dev, a.text_bytes, rank)
    nb = len(blocks)
    total = sum(n for _, n in blocks)
    caps = [(eng.block_bound(n, b"", b"") + 63) & ~63 for _, n in blocks]
    ctxs = []

    def add_ctx(e_):
        outs_ = torch.zeros(sum(caps), dtype=torch.uint8, device=dev)
        jobs_ = (E.BlockJob * nb)()
        p_out = 0
        for k, (t, n) in enumerate(blocks):
            jobs_[k].in_ = t.data_ptr(); jobs_[k].n = n; jobs_[k].method = method
            jobs_[k].filename = b""; jobs_[k].comment = b""; jobs_[k].dosha1 = 1
            jobs_[k].out = outs_.data_ptr() + p_out; jobs_[k].out_cap = caps[k]
            p_out += caps[k]
        ctxs.append((e_, jobs_, outs_))
    add_ctx(eng)
    jobs, outs = ctxs[0][1], ctxs[0][2]
    torch.cuda.synchronize()

    def step(c=0):
        e_, j_, _ = ctxs[c]
        e_.compress_blocks_dev(j_, nb)
        return sum(j_[k].out_len for k in range(nb))

    def run_steps(n):
        import threading
        done = [0.0] * n
        if depth == 1:
            r = 0
            for i in range(n):
                r = step(0); done[i] = time.perf_counter()
            return r, done
        nxt, lock, res, errs = [0], threading.Lock(), [0], []

        def worker(c, delay):
            try:
                torch.cuda.set_device(local)
                time.sleep(delay)
                while True:
                    with lock:
                        i = nxt[0]; nxt[0] += 1
                    if i >= n:
                        return
                    res[0] = step(c); done[i] = time.perf_counter()
            except Exception as ex:
                errs.append(ex)
        th = [threading.Thread(target=worker, args=(c, c * stagger / depth)) for c in range(depth)]
        for t in th: t.start()
        for t in th: t.join()
        if errs:
            raise errs[0]
        return res[0], done

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        for e_ in engines:
            e_.sync()
    # (twelve timed jobs since round 6: with three in flight a six-job window scattered by 9 % between runs of one tree; three, four
    #  and five jobs in flight are the same 670-690 ms per job: profiles/r06v_sweep_text_m2_jobs_in_flight.txt)
    steps = a.steps if a.steps is not None else 12
    warm = a.warmup if a.warmup is not None else 1
    # one job alone with one context alive (sizing step, warm step, two timed ones): the one-archive figure, and the kernel
    # durations the roofline is computed from
    step(0); step(0)
    eng.profile(not a.no_kernel_timing)
    ts, ob_ = [], 0
    for _ in range(2):
        barrier(); t_ = time.perf_counter(); ob_ = step(0); barrier(); ts.append(time.perf_counter() - t_)
    kern_alone, n_alone = eng.profile_report(), len(ts)
    eng.profile(False)
    stagger = min(ts)
    single = {"ms": round(min(ts) * 1e3, 3), "value": round(ob_ / 1e6 / min(ts), 3), "unit": "MB/s", "runs_ms": [round(t * 1e3, 1) for t in ts],
              "contexts_alive": 1, "note": "one whole job with nothing else on the chip; its floor is the SHA-1 of a 64 MiB block on one wave (~0.87 s)"}
    for c in range(1, depth):           # the other contexts; one untimed step each sizes its scratch
        engines.append(Engine(local))
        add_ctx(engines[-1])
        step(c)
    def prof_on():
        for e_ in engines:
            e_.profile(not a.no_kernel_timing)
    dt, prof_steps, out_bytes, completions = steady_window(run_steps, barrier, depth, warm, steps, prof_on)
    kern = {}
    for e_ in engines:
        for k_, (c_, m_) in e_.profile_report().items():
            kern[k_] = (kern.get(k_, (0, 0.0))[0] + c_, kern.get(k_, (0, 0.0))[1] + m_)
        e_.profile(False)
    cold = None
    if completions is not None:
        barrier(); t0 = time.perf_counter()
        out_bytes, _ = run_steps(steps)
        barrier(); cold = (time.perf_counter() - t0) / steps
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if _CPU_COLLECTIVES else dev)
        tsum = torch.tensor([float(out_bytes)], dtype=torch.float64, device="cpu" if _CPU_COLLECTIVES else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax.item()); out_bytes = int(tsum.item())
    if rank == 0:
        sec = dt / steps
        # algorithmic bytes per step (SURVEY 8d has no figure for the suffix sort; the floor of any construction is the
        # n input bytes read and the 4n bytes of suffix array written): candidates read SA/ISA/LCP once and write a decision
        # (the two candidate passes each read text + suffix array + LCP and write or update a 16-byte decision record per position)
        alg = {"sa_radix_sort_pairs": 5 * total, "lz77_sa_cand0_kernel": total * (1 + 4 + 2) + 16 * total, "lz77_sa_cand1_kernel": total * (1 + 4 + 2) + 32 * total,
               "sa_lcp_kernel": total * (1 + 4 + 4) + 2 * total, "sha1_chain_kernel": total}
        sa_ms = sum(m for k_, (c_, m) in kern.items() if k_.startswith("sa_")) / prof_steps
        # the algorithmic bytes of this path are 1 byte read + r bytes written per input byte, whatever a pass of the
        # implementation moves; `alg` above (suffix array, LCP, decision records) is the implementation's own traffic model and is
        # reported as `traffic_model`, never as `achieved`
        model = alg
        per_rank_out = out_bytes // max(1, world)
        alg = {k: total + per_rank_out for k in model}
        alg["sha1_chain_kernel"] = total

        traffic = {}
        tf = os.path.join(ROOT, "profiles", "traffic_text_m2.json")      # PMC bytes per launch (tools/gpu_traffic.sh, profiles/summarize.py)
        if os.path.exists(tf):
            tj = json.load(open(tf))
            traffic = dict(tj.get("bytes_per_launch", {}))
            # the sort scope is several rocPRIM launches: its bytes per scope = all rocPRIM bytes of the profiled jobs / jobs / scopes per job
            if tj.get("jobs_profiled") and tj.get("bytes_total", {}).get("rocprim") and "sa_radix_sort_pairs" in kern_alone:
                traffic["sa_radix_sort_pairs"] = int(tj["bytes_total"]["rocprim"] / tj["jobs_profiled"] / (kern_alone["sa_radix_sort_pairs"][0] / n_alone))

        def roof(k, src=None, nsteps=None, how=None):
            src = kern if src is None else src
            nsteps = prof_steps if nsteps is None else nsteps
            cnt, ms = src[k]
            ach = alg[k] / 1e9 / (ms / nsteps / 1e3)
            r = {"bound": "hbm", "kernel": k, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                 "traffic": traffic.get(k), "avg_launch_ms": round(ms / cnt, 4), "launches_per_step": round(cnt / nsteps, 2),
                 "algorithmic_bytes_per_step": int(alg[k]), "ms_per_step": round(ms / nsteps, 3),
                 "algorithmic": "SURVEY 8(d): 1 B read + r B written per input byte",
                 "traffic_model": {"bytes_per_step": int(model[k]), "GBps": round(model[k] / 1e9 / (ms / nsteps / 1e3), 2),
                                   "note": "what this pass of the implementation must move (text, suffix array, LCP, decision records)"}}
            if traffic.get(k):
                r["traffic_over_algorithmic"] = round(traffic[k] * (cnt / nsteps) / alg[k], 3)
            if how:
                r["measured"] = how
            return r
        # `roofline`: the chip-filling kernel with the most time in a job that ran alone; the block checksum chain (one wave per
        # 64 MiB block: what bounds one job's latency) is `roofline_longest_chain`
        how = "one job alone, %d runs (single_job)" % n_alone
        dom = max((k for k in kern_alone if k in alg and k != "sha1_chain_kernel"), key=lambda k: kern_alone[k][1], default=None)
        res = {"metric": "MB/s compressed output at -m2 (LZ77 over a suffix array), %d bytes of text in 64 MiB - 4 KiB blocks" % total,
               "value": round(out_bytes / 1e6 / sec, 3), "unit": "MB/s", "n_gpus": world, "steps": steps, "warmup": warm,
               "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": "text_m2", "input_bytes": total * world, "blocks": nb * world, "block_bytes": bs,
                          "method": "2 -> x6,1,4,0,7,27,1", "ratio": round(out_bytes / (total * world), 4)},
               "input_GBps": round(total * world / 1e9 / sec, 3), "steps_in_flight": depth, "ms_per_step_serial": round(stagger * 1e3, 3),
               "single_job": single,
               "suffix_array_ms_per_step": round(sa_ms, 2),
               "kernels_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
               **({"ms_per_step_cold": round(cold * 1e3, 3), "timing": STEADY_NOTE, "jobs_between_barriers": prof_steps} if cold else {}),
               "kernels_ms_per_job_alone": {k: round(v[1] / n_alone, 3) for k, v in sorted(kern_alone.items(), key=lambda kv: -kv[1][1])},
               "roofline": roof(dom, kern_alone, n_alone, how) if dom else None,
               "roofline_in_flight": roof(dom) if dom and dom in kern else None,
               "roofline_longest_chain": roof("sha1_chain_kernel", kern_alone, n_alone, how) if "sha1_chain_kernel" in kern_alone else None,
               **({"roofline_all": [roof(k) for k in sorted((k for k in kern if k in alg), key=lambda k: -kern[k][1])]} if a.roofline_all else {})}
        if world == 1 and not a.no_verify:
            # every block back through the device decoder (stored SHA-1 checked), bytes compared with the input
            framed, p_out = [], 0
            for k in range(nb):
                framed.append((p_out, jobs[k].out_len)); p_out += caps[k]
            uj = (E.UnblockJob * nb)()
            back = [torch.empty(n + 64, dtype=torch.uint8, device=dev) for _, n in blocks]
            for k in range(nb):
                uj[k].in_ = outs.data_ptr() + framed[k][0]; uj[k].n = framed[k][1]
                uj[k].out = back[k].data_ptr(); uj[k].out_cap = blocks[k][1] + 64
            eng.decompress_blocks_dev(uj, nb, True)
            torch.cuda.synchronize()
            bad = [(k, int(uj[k].status), int(uj[k].out_len)) for k in range(nb)
                   if not (uj[k].status == 0 and uj[k].out_len == blocks[k][1] and bool(torch.equal(back[k][: blocks[k][1]], blocks[k][0][: blocks[k][1]])))]
            res["verified_roundtrip_all_blocks"] = not bad
            if bad:
                res["roundtrip_failures"] = bad[:8]
            del back
        if world == 1 and not a.no_cpu_baseline:
            # the real reference (divsufsort + LZBuffer) over the same blocks on the host cores, whole job; it also
            # compares every code stream with the one inside the GPU's framed block
            blob = b"".join(bytes(t[:n].cpu().numpy()) for t, n in blocks)
            fr = b"".join(bytes(outs[o:o + l].cpu().numpy()) for o, l in ((sum(caps[:k]), jobs[k].out_len) for k in range(nb)))
            lens = [jobs[k].out_len for k in range(nb)]
            cb = cpu_baseline("m2", [bs, json.dumps(lens)], [blob, fr])
            if "identical_blocks" in cb:
                res["verified_blocks"] = cb.pop("identical_blocks")
                res["verified_all_blocks"] = res["verified_blocks"] == "%d of %d" % (nb, nb)
            res["cpu_baseline"] = cb
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()
    for e_ in engines:
        e_.close()


def main_cm_m5(a, rank, world, local, dev):
    """Context-mixing stress ("-m3/-m4 runs to stress context mixing"; north_star: one wavefront per
    ZPAQ block, thousands of blocks in flight): `--cm-blocks` blocks of `--cm-block-bytes` of text per GPU through
    compressBlock at level 5 -- Predictor (22 components: ICM/ISSE chains, MATCH, three mixers, SSE) + arithmetic coder +
    HCOMP per byte, one wave per block, all blocks resident at once (90 MB of model state each).  Blocks are independent:
    no collective on the data path; weak scaling."""
    from zpaqfranz_amd import Engine, engine as E
    bs = a.cm_block_bytes
    # blocks resident at once: as many as HBM holds (84 MiB of model state each at this block size) -- 3072 = three waves per SIMD
    # where they fit, else fewer (round 6: 2048 / 2560 blocks = 113 / 125 MB/s in, profiles/r06s_cm_blocks_resident.txt)
    candidates = [a.cm_blocks] if a.cm_blocks else [3072, 2816, 2560, 2048]
    eng = Engine(local)
    blocks_all = text_blocks_dev(dev, candidates[0] * bs, rank, block=bs)
    assert len(blocks_all) == candidates[0]
    first = bytes(blocks_all[0][0][: blocks_all[0][1]].cpu().numpy())
    xm = E.expand_method("50", first)
    method = xm.encode()
    torch.cuda.empty_cache()
    nb = 0
    for cand in candidates:
        probe = (E.BlockJob * cand)()
        cap1 = (eng.block_bound(bs, b"", b"") + 63) & ~63
        scratch = torch.empty(cand * cap1, dtype=torch.uint8, device=dev)
        for k, (t, n) in enumerate(blocks_all[:cand]):
            probe[k].in_ = t.data_ptr(); probe[k].n = n; probe[k].method = method
            probe[k].filename = b""; probe[k].comment = b""; probe[k].dosha1 = 1
            probe[k].out = scratch.data_ptr() + k * cap1; probe[k].out_cap = cap1
        try:
            eng.compress_blocks_dev(probe, cand)            # (the sizing step: the model arena of this many blocks)
            ok = all(probe[k].status == 0 for k in range(cand))
        except E.ZpqError as ex:
            ok = False
            if ex.status != -8:                             # anything but ZPQ_ERR_NOMEM is an error
                raise
        del scratch
        torch.cuda.empty_cache()
        if ok:
            nb = cand
            break
        eng.close(); eng = Engine(local)                     # (the context that failed keeps nothing)
    if not nb:
        raise SystemExit("cm_m5: not even %d blocks fit" % candidates[-1])
    blocks = blocks_all[:nb]
    del blocks_all
    total = sum(n for _, n in blocks)
    # what compressBlock's "50" (level 5, blocks up to 1 MiB) expands to for this data (level 5 looks at the data to add
    # models for periodic structure: text has none, checked for every block outside the timed region); passing the
    # expansion itself keeps that host-side analysis out of the measured step
    src, args = E.make_config(xm)
    header = E.compile_config(src, args)[0]
    caps = [(eng.block_bound(n, b"", b"") + 63) & ~63 for _, n in blocks]
    outs = torch.zeros(sum(caps), dtype=torch.uint8, device=dev)
    jobs = (E.BlockJob * nb)()
    p_out, offs = 0, []
    for k, (t, n) in enumerate(blocks):
        jobs[k].in_ = t.data_ptr(); jobs[k].n = n; jobs[k].method = method
        jobs[k].filename = b""; jobs[k].comment = b""; jobs[k].dosha1 = 1
        jobs[k].out = outs.data_ptr() + p_out; jobs[k].out_cap = caps[k]
        offs.append(p_out); p_out += caps[k]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    def step():
        eng.compress_blocks_dev(jobs, nb)
        bad = [k for k in range(nb) if jobs[k].status != 0]
        if bad:
            raise RuntimeError("compressBlock failed for block %d: status %d" % (bad[0], jobs[bad[0]].status))
        return sum(jobs[k].out_len for k in range(nb))

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()
    steps = a.steps if a.steps is not None else 2
    warm = a.warmup if a.warmup is not None else 1
    for _ in range(warm):
        step()
    eng.profile(not a.no_kernel_timing)
    barrier()
    t0 = time.perf_counter()
    out_bytes = 0
    for _ in range(steps):
        out_bytes = step()
    barrier()
    dt = time.perf_counter() - t0
    kern = eng.profile_report()
    eng.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if _CPU_COLLECTIVES else dev)
        tsum = torch.tensor([float(out_bytes)], dtype=torch.float64, device="cpu" if _CPU_COLLECTIVES else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax.item()); out_bytes = int(tsum.item())
    if rank == 0:
        sec = dt / steps
        k_ms = kern.get("cm_spec_encode", kern.get("cm_wave_kernel", (1, 0.0)))
        per = k_ms[1] / max(1, k_ms[0])
        ach = (total + out_bytes / world) / 1e9 / (per / 1e3) if per else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_cm_m5.json")       # PMC bytes per launch of the 2048 x 64 KiB profile run, scaled by input size
        if os.path.exists(tf):
            t_ = json.load(open(tf)).get("bytes_per_launch", {}).get("cm_spec_encode")
            if t_:
                traffic = int(t_ * total / (2048.0 * 65536))
        res = {"metric": "MB/s compressed output at -m5 (context mixing: 22-component Predictor + arithmetic coder), %d blocks of %d KiB of text per GPU" % (nb, bs >> 10),
               "value": round(out_bytes / 1e6 / sec, 3), "unit": "MB/s", "n_gpus": world, "steps": steps, "warmup": warm,
               "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": "cm_m5", "input_bytes": total * world, "blocks": nb * world, "block_bytes": bs,
                          "method": "50 -> " + xm, "components": header[6], "ratio": round(out_bytes / (total * world), 4)},
               "input_MBps": round(total * world / 1e6 / sec, 3),
               "blocks_in_flight_per_gpu": nb, "waves_per_simd": round(nb / 1024.0, 2),
               "input_KBps_per_block": round(total / nb / 1e3 / (per / 1e3), 1) if per else None,
               "kernels_ms_per_step": {k: round(v[1] / steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
               "roofline": {"bound": "hbm", "kernel": "cm_spec_encode", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic, "avg_launch_ms": round(per, 3),
                            "launches_per_step": round(k_ms[0] / steps, 2), "algorithmic_bytes_per_step": int(total + out_bytes / world),
                            "note": "one serial chain of bit decisions per block (a wave each): bounded by instruction issue and dependent "
                                    "table lookups, not by HBM bytes (1 B read + r B written per input byte)"}}
        if world == 1 and not a.no_verify:
            # (1) every framed block back through the device decoder (context-model decode + stored SHA-1), bytes compared
            uj = (E.UnblockJob * nb)()
            back = torch.zeros(nb * (bs + 64), dtype=torch.uint8, device=dev)
            for k in range(nb):
                uj[k].in_ = outs.data_ptr() + offs[k]; uj[k].n = jobs[k].out_len
                uj[k].out = back.data_ptr() + k * (bs + 64); uj[k].out_cap = blocks[k][1] + 64
            t1 = time.perf_counter()
            eng.decompress_blocks_dev(uj, nb, True)
            torch.cuda.synchronize()
            res["decode_ms"] = round((time.perf_counter() - t1) * 1e3, 1)
            bad = [(k, int(uj[k].status)) for k in range(nb)
                   if not (uj[k].status == 0 and uj[k].out_len == blocks[k][1] and
                           bool(torch.equal(back[k * (bs + 64): k * (bs + 64) + blocks[k][1]], blocks[k][0][: blocks[k][1]])))]
            res["verified_roundtrip_all_blocks"] = not bad
            if bad:
                res["roundtrip_failures"] = bad[:8]
            del back
            # (2) the expansion of "50" is the same for every block (no periodic models found anywhere)
            same = all(E.expand_method("50", bytes(t[:n].cpu().numpy())) == xm for t, n in blocks[:: max(1, nb // 64)])
            res["method_expansion_checked"] = bool(same)
        if world == 1 and not a.no_cpu_baseline:
            # the reference Predictor (x86 JIT) + mirrored Encoder on every host core over a bounded sample of the same
            # blocks; every sampled code stream is compared with the one inside the GPU's framed block
            sample = min(nb, a.cm_cpu_sample)
            blob = b"".join(bytes(t[:n].cpu().numpy()) for t, n in blocks[:sample])
            fr = b"".join(bytes(outs[offs[k]: offs[k] + jobs[k].out_len].cpu().numpy()) for k in range(sample))
            cb = cpu_baseline("cm", [bs, json.dumps([jobs[k].out_len for k in range(sample)]), header.hex(), nb], [blob, fr])
            if "identical_blocks" in cb:
                res["verified_blocks_vs_reference_coder"] = cb.pop("identical_blocks")
                res["verified_all_sampled_blocks"] = res["verified_blocks_vs_reference_coder"] == "%d of %d" % (sample, sample)
            res["cpu_baseline"] = cb
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()
    eng.close()


def compact_line(d, top=6):
    """A nested workload's JSON line cut down to what a reader of the ONE line needs (the driver's record keeps only the
    tail of a long line): the contract keys, the verification flags, the dominant kernel's roofline and the CPU baseline.
    The full line goes to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "ms_per_step_serial", "steps_in_flight", "scaling", "dtype",
            "data", "input_GBps", "output_GBps", "input_MBps", "sha256_mismatches", "error", "rc", "skipped", "wall_s")
    out = {k: d[k] for k in keep if k in d}
    out.update({k: v for k, v in d.items() if k.startswith("verified") or k.endswith("_failures") or k == "method_expansion_checked"})
    if "config" in d:
        out["config"] = {k: v for k, v in d["config"].items() if k in ("workload", "input_bytes", "files", "blocks", "method", "ratio", "unique_bytes", "out_bytes")}
    for rk in ("roofline", "roofline_in_flight", "roofline_longest_chain"):
        r = d.get(rk)
        if r:
            out[rk] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_ms", "launches_per_step", "ms_per_step", "waves", "measured") if k in r}
    if d.get("roofline_end_to_end"):
        out["roofline_end_to_end"] = {k: d["roofline_end_to_end"][k] for k in ("achieved", "frac", "algorithmic_bytes_per_step")}
    if d.get("kernels_ms_per_step"):
        out["kernels_ms_per_step"] = dict(list(d["kernels_ms_per_step"].items())[:top])
    cb = d.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cb}
    for k in ("blake3_verify", "no_fold", "every_byte_hashed", "twin_fold_on", "single_job", "product_one_call", "archive_blocks"):
        if isinstance(d.get(k), dict):
            out[k] = {kk: vv for kk, vv in d[k].items() if kk != "note"}
    return out


def summary_row(d):
    """[value, ms_per_step, verified, roofline.frac, cpu_baseline.value] of one workload line"""
    if "value" not in d:
        return [None, None, False, None, None]
    flags = [v for k, v in d.items() if k.startswith("verified") and isinstance(v, bool)]
    return [d.get("value"), d.get("ms_per_step"), bool(flags) and all(flags), (d.get("roofline") or {}).get("frac"),
            (d.get("cpu_baseline") or {}).get("value")]


def run_other_workloads(a):
    """dup8_m1 (config 4 at 1-GPU size), extract_m1 (config 5), text_m2 (config 3) and cm_m5 (context-mixing stress), each
    as `bench.py --workload X` in a fresh process with its default size; returns {workload: its JSON line or an error}."""
    import subprocess
    out = {}
    budget = float(os.environ.get("ZPQ_BENCH_OTHERS_S", "1200"))
    t_all = time.time()
    for w in ("extract_m1", "text_m2", "cm_m5", "dup8_m1"):
        left = budget - (time.time() - t_all)
        if left < 60:
            out[w] = {"skipped": "time budget for the nested workloads used up"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--gpus", "1"]
        for flag, on in (("--no-cpu-baseline", a.no_cpu_baseline), ("--no-verify", a.no_verify)):
            if on:
                cmd.append(flag)
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                out[w] = json.loads(lines[-1])
                out[w]["wall_s"] = round(time.time() - t0, 1)
            else:
                out[w] = {"error": (r.stderr or r.stdout)[-400:], "rc": r.returncode}
        except subprocess.TimeoutExpired:
            out[w] = {"error": "timed out after %.0f s" % left}
    return out


def self_launch(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (what the driver's torch.distributed.run
    command line does), pass their output through and leave with their return code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def cpu_baseline(mode, argv, blobs):
    """Runs tests/cpu_baseline.py (the reference's own code on all host cores) in a fresh process and returns its JSON."""
    import subprocess
    import tempfile
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    files = []
    try:
        for blob in blobs:
            f = tempfile.NamedTemporaryFile(dir=d, suffix=".bin", delete=False)
            f.write(blob); f.close(); files.append(f.name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cpu_baseline.py"), mode] + files + [str(a) for a in argv],
                           capture_output=True, text=True, timeout=900)
    finally:
        for fn in files:
            try: os.unlink(fn)
            except OSError: pass
    if r.returncode != 0:
        return {"value": None, "error": r.stderr[-300:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=["all", "silesia_x256_m1", "dup8_m1", "extract_m1", "text_m2", "cm_m5"],
                    help="default: silesia_x256_m1 (BASELINE config 2) as the headline, then -- single GPU only -- every other workload "
                         "in its own process, nested under 'workloads' in the one JSON line")
    ap.add_argument("--cm-blocks", type=int, default=None, help="cm_m5: blocks per GPU (one wave each, all resident at once); default: 3072 / 2816 / 2560 / 2048, the most that fit HBM")
    ap.add_argument("--cm-block-bytes", type=int, default=256 << 10, help="cm_m5: bytes per block")
    ap.add_argument("--cm-cpu-sample", type=int, default=192, help="cm_m5: blocks the CPU baseline codes (and compares)")
    ap.add_argument("--text-bytes", type=int, default=10 ** 9, help="text_m2: bytes of text per GPU (enwik9 is 10^9)")
    ap.add_argument("--copies", type=int, default=256, help="corpus replication factor (256 = BASELINE config)")
    ap.add_argument("--units", type=int, default=1024, help="dup8_m1: unique 16 MiB units per GPU")
    ap.add_argument("--dup", type=int, default=8, help="dup8_m1: copies of every unit")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every corpus member (debug only)")
    ap.add_argument("--pipeline", type=int, default=None, help="steps in flight (each on its own engine context); 1 = strictly serial")
    ap.add_argument("--shared-corpus", action="store_true",
                    help="multi-GPU, strong scaling (the DEFAULT with more than one rank: BASELINE's metric is ONE Silesia x256 on 1/2/4/8 "
                         "GPUs): one corpus split by file range across the ranks (rank r holds copies [r*copies/N, (r+1)*copies/N)): every "
                         "fragment of ranks > 0 duplicates one of rank 0, the global first-occurrence exchange carries the whole dedup; the "
                         "stitched archive equals the single-GPU one")
    ap.add_argument("--own-corpus", action="store_true",
                    help="multi-GPU, weak scaling: every rank owns a corpus of its own (different seed), N times the single-GPU job")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="test only (runs without a GPU): the ranks meet over gloo, agree on the world size and rank 0 prints a line with n_gpus")
    ap.add_argument("--roofline-all", action="store_true", help="also print the roofline block of every timed kernel (long)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL); gloo runs the collectives through the host: functional test only")
    ap.add_argument("--same-device", action="store_true", help="test only: every rank uses GPU 0 (with --dist-backend gloo)")
    ap.add_argument("--force-collectives", action="store_true", help="single rank: run the multi-rank code path (RCCL all-gathers with world size 1)")
    ap.add_argument("--dump-archive", default=None, help="test only: rank 0 writes the stitched d blocks of the last step to this file")
    ap.add_argument("--python-pipeline", action="store_true",
                    help="add workloads, one rank: time the call-by-call orchestration in Python (fragment -> dedup -> plan -> gather -> "
                         "compressBlock, d blocks only) instead of the product's one-call zpqj_add_dev (the default: whole archive incl. c/h/i)")
    ap.add_argument("--product", action="store_true", help="dup8_m1: time zpqj_add_dev (archive to host memory) instead of the call-by-call path")
    ap.add_argument("--strict", action="store_true", help="exit with status 4 when a nested workload failed or was skipped (the line is printed first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block-sha1", action="store_true", help="experiment only: skip the per-block SHA-1 (invalid as a result)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with hipEvents (roofline block is then empty)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run bit-identity check against the oracle")
    ap.add_argument("--twins", action="store_true",
                    help="timed region WITH the twin-file fold (files whose bytes equal an earlier file's are found by comparison and take its "
                         "fragment records).  Default since round 5: fold OFF in the timed region -- every byte goes through the fragment loop "
                         "and SHA-1, as in Jidac::add -- and the fold's figure is reported beside it as 'twin_fold_on' / 'value_twin_fold'")
    ap.add_argument("--no-twins", action="store_true", help="(the default; kept for older command lines)")
    a = ap.parse_args()
    run_all = a.workload in (None, "all")
    if run_all:
        a.workload = "silesia_x256_m1"
    if os.environ.get("ZPQ_BENCH_WATCHDOG"):       # debugging aid: dump every thread's stack and exit if the run takes too long
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ZPQ_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)                 # N ranks under torch.distributed.run; does not return
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): the line would carry the wrong n_gpus" % (a.gpus, world))
    if a.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
        t = torch.tensor([1 << rank], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "ranks_seen": int(t.item()), "rendezvous_only": True,
                              "scaling": "weak" if a.own_corpus else "strong"}))
        dist.destroy_process_group()
        return
    if run_all and world == 1 and not a.force_collectives:
        # every workload in its own process (its own HIP context: what one leaves in HBM never shrinks the batches of the
        # next), this process only stitches their JSON lines: the headline stays config 2, the rest nests under "workloads"
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "silesia_x256_m1", "--gpus", "1"]
        for flag, v in (("--steps", a.steps), ("--warmup", a.warmup), ("--pipeline", a.pipeline)):
            if v is not None:
                cmd += [flag, str(v)]
        if a.copies != 256:
            cmd += ["--copies", str(a.copies)]
        if a.scale != 1.0:
            cmd += ["--scale", str(a.scale)]
        for flag, on in (("--no-cpu-baseline", a.no_cpu_baseline), ("--no-verify", a.no_verify), ("--no-kernel-timing", a.no_kernel_timing),
                         ("--twins", a.twins)):
            if on:
                cmd.append(flag)
        r = subprocess.run(cmd, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit("the headline workload failed (rc %d)" % r.returncode)
        res = json.loads(lines[-1])
        others = run_other_workloads(a)
        for w, d in others.items():                     # the full lines: stderr (the ONE stdout line stays short enough to be read whole)
            sys.stderr.write("[bench.py] %s: %s\n" % (w, json.dumps(d)))
        # order matters to a reader who only sees the END of the line: details first, then the headline's own roofline /
        # cpu_baseline, the every-byte-hashed figure and one row per workload
        tail = {k: res.pop(k) for k in ("roofline_in_flight", "roofline_longest_chain", "roofline_end_to_end", "roofline", "cpu_baseline",
                                         "single_job", "every_byte_hashed", "twin_fold_on", "value_twin_fold") if k in res}
        res["workloads"] = {w: compact_line(d) for w, d in others.items()}
        res.update(tail)
        if "every_byte_hashed" in res:
            res["value_every_byte_hashed"] = res["every_byte_hashed"]["value"]
        res["workloads_summary"] = {"columns": ["value", "ms_per_step", "verified", "roofline.frac", "cpu_baseline.value"],
                                    res["config"]["workload"]: summary_row(res), **{w: summary_row(d) for w, d in others.items()}}
        failed = [w for w, d in others.items() if "value" not in d]
        if failed:
            res["failed_workloads"] = failed
        print(json.dumps(res))
        if failed:           # reported in the line (failed_workloads, workloads_summary); the headline itself is complete
            sys.stderr.write("bench.py: nested workload(s) failed or were skipped: " + ", ".join(failed) + "\n")
            if a.strict or os.environ.get("ZPQ_BENCH_STRICT"):
                raise SystemExit(4)          # (CI: a failed nested workload fails the run; the driver's run keeps rc 0 and reads failed_workloads)
        return
    if a.same_device:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or a.force_collectives:
        global _CPU_COLLECTIVES
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.dist_backend == "gloo":
            _CPU_COLLECTIVES = True
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if a.workload == "text_m2":
        return main_text_m2(a, rank, world, local, dev)
    if a.workload == "cm_m5":
        return main_cm_m5(a, rank, world, local, dev)
    # (extract_m1: twelve timed jobs since round 6 -- with four in flight and six in the window two default runs of one tree gave
    #  724 and 923 ms per step, profiles/r06d_bench*.json)
    steps = a.steps if a.steps is not None else {"silesia_x256_m1": 48, "dup8_m1": 2, "extract_m1": 12}[a.workload]
    warm = a.warmup if a.warmup is not None else {"silesia_x256_m1": 3, "dup8_m1": 1, "extract_m1": 2}[a.workload]
    # steps in flight: the add path hides its serial tails (216 ms of block checksum chain, ~140 ms of LZ77 parse per 2 MiB
    # segment) behind the chip-wide kernels of other steps; round 4 (three-wave parse, 2 MiB segments: 12.5 GB of table states
    # per job): 106.8 / 99.5 / 98.0 ms per step at 8 / 10 / 12, 133 at 14 (the tables no longer fit): twelve by default until round 6
    # (multi-rank runs too: every step in flight adds three collective sections to the one fixed order, CollectiveOrder)
    multi = world > 1 or a.force_collectives
    # (round 6, steady-state window, three interleaved runs each -- profiles/r06k_sweep_jobs_in_flight_steady_state.txt: 9 / 10 / 11 / 12 in
    #  flight = 97.4 / 92.5 / 90.9 / 99.4 ms per step: eleven)
    # (extract_m1: five since round 6 -- the MI355X's 288 GiB hold five sets of restored files, 54 GB each; 757 / 711 ms per job at four in
    #  flight, 627 at five: profiles/r06m_extract_chain_priority_and_five_in_flight.txt; capped by free HBM below)
    depth = max(1, a.pipeline if a.pipeline is not None else {"silesia_x256_m1": 11, "dup8_m1": 1, "extract_m1": 5}[a.workload])
    import datagen
    from zpaqfranz_amd import Engine
    eng = Engine(local)
    shared = a.workload == "silesia_x256_m1" and (a.shared_corpus or (world > 1 and not a.own_corpus))
    corpus = datagen.silesia_like(seed=0 if shared else rank, scale=a.scale)
    if a.workload == "dup8_m1":
        layout = dup8_layout(dev, corpus, a.units, a.dup, rank)
    elif shared:
        if a.copies % world:
            raise SystemExit("--shared-corpus: --copies must be a multiple of the number of ranks")
        layout = silesia_layout(dev, corpus, a.copies // world)        # this rank's file range of the one corpus
    else:
        layout = silesia_layout(dev, corpus, a.copies)
    torch.cuda.empty_cache()       # what the generators left in torch's cache is HBM the engine cannot see (its LZ77 batches are sized by free memory)
    if a.workload == "silesia_x256_m1" and a.pipeline is None:
        # every job in flight holds its own hash-table states (12.5 GB for the 13 blocks at 2 MiB segments) and ~3 GB of
        # fragment tables, block buffers and outputs: no deeper than HBM allows
        free_b = torch.cuda.mem_get_info(dev)[0]
        depth = max(1, min(depth, int((free_b - (24 << 30)) // (16 << 30))))
    # `pipeline` steps in flight on as many engine contexts and threads; with several ranks the collectives of the
    # steps in flight are issued in one fixed order on every rank (CollectiveOrder)
    # ... created AFTER one job has been timed with a single context alive (`single_job`: idle contexts oversubscribe the
    # hardware queues and cost a lone job what they give the twelve)
    engines, pipes = [eng], []
    # one rank: the timed step is ONE C-ABI call of the product, zpqj_add_dev (files resident in HBM in, c/d/h/i archive out)
    # (config 4's archive is 5.7 GB: returned to host memory it turns the step into a PCIe / host-copy measurement, so its timed step stays
    #  the call-by-call one with the d blocks left in HBM; ONE product call is made and verified after the timed region: `product_one_call`)
    product = (a.workload == "silesia_x256_m1" or (a.workload == "dup8_m1" and a.product)) and world == 1 and not a.force_collectives and not a.python_pipeline
    # several ranks (or --force-collectives): ONE call of the product per rank and step as well -- zpqj_add_sharded_dev, with the
    # in-tree RCCL collectives inside it (gloo runs, tests only, hand it functional stand-ins over torch.distributed)
    sharded_product = a.workload == "silesia_x256_m1" and multi and not a.python_pipeline
    shard_names = shard_sizes = shard_gather = None
    if sharded_product:
        # every file of the job, in the order the ranks hold them (rank r: copies [r C, (r + 1) C) of its corpus); names ascend
        cps = layout["copies"]
        # (one corpus over the ranks: the names one GPU would use for the whole of it -- the archive is then the single-GPU archive)
        shard_names = [("c%04d/%s" % (r_ * cps + c_, n_)) if shared else ("r%02d/c%04d/%s" % (r_, c_, n_))
                       for r_ in range(world) for c_ in range(cps) for n_ in layout["names"]]
        shard_sizes = [int(z_) for _ in range(world) for _ in range(cps) for z_ in layout["sizes"]]
        assert shard_names == sorted(shard_names), "the corpus members must be held in name order"
    # RCCL: every add in flight has a communicator of its own (created in the same order on every rank) and the adds are dealt to
    # the contexts statically -- job i runs on context i mod depth on every rank -- so each communicator sees the same sequence of
    # calls everywhere and no order between the adds in flight has to be imposed: the C collectives go to the product call as
    # function pointers, nothing of Python or torch stands between the engine and RCCL.  (With ONE communicator and the fixed
    # turn order of CollectiveOrder every RCCL launch of every add queued behind the others, and each waits for a free slot on a
    # chip that is full of chip-filling kernels: 217-221 ms per step at world size 1 against 99 ms without collectives,
    # profiles/r06e_rccl_world1_own_stream.txt.)  gloo (tests) keeps the one group and the turn order.
    comm_per_add = sharded_product and a.dist_backend != "gloo"
    gathers = []

    def new_gather(e_):
        from zpaqfranz_amd import engine as E_
        uid = [E_.RcclGather.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        g_ = E_.RcclGather(e_, rank, world, uid[0])
        gathers.append(g_)
        return g_

    def add_pipe(e_):
        p_ = Pipeline(e_, dev, layout, rank, world, a.force_collectives)
        p_.no_block_sha1 = a.no_block_sha1
        p_.use_twins = a.twins
        p_.balance_blocks = shared and world > 1      # one corpus over several ranks: the d blocks are dealt out, not left to rank 0
        p_.product = product
        if sharded_product:
            from zpaqfranz_amd import engine as E_
            if comm_per_add:
                p_.use_sharded_product(shard_names, shard_sizes, new_gather(e_), None)
            else:
                p_.use_sharded_product(shard_names, shard_sizes, E_.dist_allgather_bytes(), E_.dist_allgather_dev(e_))
        pipes.append(p_)
        return p_
    add_pipe(eng)
    ex_pipe = None
    if a.workload == "extract_m1":
        import hashlib
        if world > 1:
            raise SystemExit("extract_m1: blocks are independent -- run N single-GPU replicas (no collective on this path)")
        pipes[0].step()                                     # the archive to extract (untimed)
        sha = [hashlib.sha256(b).digest() for _, b in corpus]
        ex_pipe = ExtractPipeline(eng, dev, pipes[0], layout, sha * a.copies)
        # since round 6 the timed region hashes EVERY restored byte (fold off, as the add headline since round 5); the fold's figure
        # is reported beside it (`twin_fold_on`); --twins times the fold
        ex_pipe.use_twins = a.twins
        if not a.python_pipeline:
            pipes[0].step_product(True)                     # the same job through zpqj_add_dev: the whole archive (c, d, h, i blocks)
            ex_pipe.use_product(bytes(pipes[0].archive))
            pipes[0].drop_archive()
        layout["data"] = None; del pipes[0].verify_blocks  # the originals are not needed any more
        for p_ in pipes:
            p_.data = None
        torch.cuda.empty_cache()
        # several extract jobs in flight (own context, own output): a job is as long as its longest serial chain -- the
        # SHA-256 of the 51 MB member on one wave, the SHA-1 of a 16 MiB block -- and leaves most of the chip idle
        runners = [ex_pipe]
        # the context that wrote the archive keeps ~20 GB of grow-only scratch (the LZ77 table states of the add): the extract jobs
        # get fresh contexts, that one is closed
        eng.close()
        eng = Engine(local)
        engines[0] = eng
        ex_pipe.eng = eng
        torch.cuda.empty_cache()
        if a.pipeline is None:
            # every further job in flight holds its own restored files and decoded blocks; 12 GB stay free for the scratch arenas
            # of the contexts (BLAKE3 chaining values, LZ77 decode records) and the verification's copies
            per_job = ex_pipe.out.numel() + ex_pipe.plain.numel() + (1 << 28)
            free_b = torch.cuda.mem_get_info(dev)[0]
            depth = max(1, min(depth, 1 + int((free_b - (12 << 30)) // per_job)))
    else:
        runners = pipes

    def more_contexts():
        for _ in range(1, depth):
            engines.append(Engine(local))
            if ex_pipe is not None:
                runners.append(ex_pipe.clone_for(engines[-1]))
            else:
                add_pipe(engines[-1])

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        for e_ in engines:
            e_.sync()

    def run_steps(n):
        """n whole-job steps; with depth > 1 they are software-pipelined: while one step sits in its
        latency-bound tail (LZ77 parse, block checksums: a few hundred waves) the next one already
        fragments and hashes on the rest of the chip.  Every step is complete when this returns."""
        import threading
        nxt, lock, outs, errs, done = [0], threading.Lock(), [0] * n, [], [0.0] * n

        order = (CollectiveOrder(n, depth, Pipeline.SECTIONS if sharded_product else 3)
                 if ((world > 1 or a.force_collectives) and not comm_per_add) else _NoOrder())
        nworkers = 1 if (depth == 1 or len(runners) == 1) else len(runners)

        def worker(p_, delay, wk=0):
            try:
                torch.cuda.set_device(local)     # the current device is per host thread
                time.sleep(delay)      # stagger: one step's chip-wide kernels against the other's latency-bound tail
                mine = wk
                while True:
                    if comm_per_add:       # static: job i on context i mod depth, on every rank alike
                        i = mine; mine += nworkers
                    else:
                        with lock:
                            i = nxt[0]; nxt[0] += 1
                    if i >= n:
                        return
                    outs[i] = p_.step(order, i, i == n - 1) if isinstance(p_, Pipeline) else p_.step(order, i)
                    done[i] = time.perf_counter()
                    if i == n - 1:
                        last_pipe[0] = p_
            except Exception as ex:       # surface worker failures in the main thread
                errs.append(ex)
                if hasattr(order, "abort"):
                    order.abort()
                if world > 1:              # a rank that stops would leave the others waiting in a collective
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
                    os._exit(3)
        if depth == 1 or len(runners) == 1:
            worker(runners[0], 0.0)
        else:
            # the jobs start `spacing` apart: single / depth by default; ZPQ_BENCH_SPACING_MS: experiment (the steady period, so that the
            # pipeline starts spread over one in-flight latency instead of bunched)
            spacing = float(os.environ["ZPQ_BENCH_SPACING_MS"]) / 1e3 if os.environ.get("ZPQ_BENCH_SPACING_MS") else stagger[0] / depth
            th = [threading.Thread(target=worker, args=(p_, k_ * spacing, k_)) for k_, p_ in enumerate(runners)]
            for t in th: t.start()
            for t in th: t.join()
        if errs:
            raise errs[0]
        return (outs[-1] if n else 0), done

    stagger = [0.0]
    last_pipe = [runners[0]]     # the context that ran the last step (its results are the ones dumped / verified)
    single, kern_alone, n_alone = None, {}, 0
    if depth > 1:
        # ONE JOB ALONE, one context alive: the one-archive figure (ZSFX/zsfx.cpp:2147-2148: one process, one archive) and the
        # kernel durations `roofline` is computed from (a kernel's time beside eleven other jobs is a latency under
        # contention, not what the kernel does on the chip).  Sizing step, warm step, then two timed ones.
        one = (lambda: runners[0].step(None, 0, False)) if ex_pipe is None else (lambda: runners[0].step())
        one(); one()
        eng.profile(not a.no_kernel_timing)
        ts, ob_ = [], 0
        for _ in range(2):
            barrier(); t_ = time.perf_counter(); ob_ = one(); barrier(); ts.append(time.perf_counter() - t_)
        kern_alone, n_alone = eng.profile_report(), len(ts)
        eng.profile(False)
        stagger[0] = min(ts)
        single = {"ms": round(min(ts) * 1e3, 3), "value": round(ob_ / 1e6 / min(ts), 3), "unit": "MB/s", "runs_ms": [round(t * 1e3, 1) for t in ts],
                  "contexts_alive": 1,
                  "note": "one whole job with nothing else on the chip: the one-archive latency; its floor is the longest serial chain "
                          "(add: SHA-1 of a 16 MiB d block on one wave, 406 instructions per 64 bytes at 4 cycles = ~215 ms; extract: "
                          "SHA-256 of the longest file)"}
        more_contexts()
        for p_ in runners[1:]:         # one untimed serial step per further context sizes its scratch
            p_.step()
    def prof_on():
        for e_ in engines:
            e_.profile(not a.no_kernel_timing)
    dt, prof_steps, out_bytes, completions = steady_window(run_steps, barrier, depth if len(runners) > 1 else 1, warm, steps, prof_on)
    kern = {}
    for e_ in engines:
        for k_, (c_, m_) in e_.profile_report().items():
            kern[k_] = (kern.get(k_, (0, 0.0))[0] + c_, kern.get(k_, (0, 0.0))[1] + m_)
        e_.profile(False)
    cold = None
    if completions is not None:
        # the same K steps from an idle chip to an idle chip (what rounds 1-5 reported as ms_per_step): fill and drain included
        barrier(); t0 = time.perf_counter()
        out_bytes, _ = run_steps(steps)
        barrier(); cold = (time.perf_counter() - t0) / steps
    pipe = last_pipe[0]
    product_stats, product_archive, product_once = None, None, None
    if (not product and a.workload == "dup8_m1" and world == 1 and not a.force_collectives and not a.python_pipeline and not a.no_verify
            and isinstance(pipe, Pipeline)):
        # ONE call of the product's zpqj_add_dev over the same files (outside the timed region): its archive -- in host memory -- is verified
        # below like the headline's, and its wall time is reported as what it is
        # (room first: the call-by-call step's block buffers and outputs -- 35 GB at this size -- go back to the driver)
        for p_ in pipes:
            p_.verify_blocks = None
        torch.cuda.empty_cache()
        try:
            t_ = time.perf_counter()
            n_ = pipe.step_product(True)
            product_once = {"ms": round((time.perf_counter() - t_) * 1e3, 1), "archive_bytes": n_,
                            "note": "zpqj_add_dev once: the same job with the whole archive (c, d, h, i blocks) returned to HOST memory -- PCIe and host copies of %.1f GB included" % (n_ / 1e9)}
            product_stats, product_archive = dict(pipe.stats), pipe.archive
        except Exception as ex:                        # (reported, not fatal: the timed region above is complete)
            product_once = {"error": str(ex)[-300:]}
        pipe.step()                                   # (the tables both verifications compare with: the call-by-call step again)
    if product and isinstance(pipe, Pipeline):
        # the same job once more through the call-by-call orchestration (untimed): its tables and d blocks are what verify_add
        # checks against the oracle, and what the archive the product returned in the timed region is compared with
        product_stats, product_archive = dict(pipe.stats), pipe.archive
        pipe.product = False
        pipe.step()
        pipe.product = True
        pipe.stats = dict(pipe.stats, out_bytes=product_stats["out_bytes"], d_bytes=product_stats["d_bytes"])
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if _CPU_COLLECTIVES else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        sec = dt / steps
        st = pipe.stats
        extract = a.workload == "extract_m1"
        if extract:
            in_bytes = pipe.total                       # restored bytes
            ub = sum(pipe.usize)
            # per STEP: the two SHA-256 kernels split the files between them (wave-wide for the longest, lane-wise for the rest)
            chain_bytes = sum(sorted((pipe.file_off[i + 1] - pipe.file_off[i] for i in range(pipe.nfiles)), reverse=True)[:1024])
            tw = getattr(pipe, "twin_stats", None) if getattr(pipe, "use_twins", True) else None
            tw = tw or dict(twins=0, twin_bytes=0, compared=0, compared_bytes=0)
            hashed = pipe.total - tw["twin_bytes"]      # bytes SHA-256 actually sees (the representatives)
            chain_bytes = min(chain_bytes, hashed)
            long_bytes = sum(n_ for n_ in (pipe.file_off[i + 1] - pipe.file_off[i] for i in range(pipe.nfiles)) if n_ >= (1 << 20))
            grouped = "sha256_group_kernel" in kern or "sha256_group_kernel" in kern_alone       # (several chains per wave: every file of 1 MiB and more)
            if grouped:
                chain_bytes = min(long_bytes, hashed)
            alg = {"sha256_chain_kernel": chain_bytes, "sha256_group_kernel": chain_bytes, "sha256_extents_kernel": hashed - chain_bytes, "gather_kernel": pipe.total + ub,
                   "twin_compare_kernel": tw["compared_bytes"] + (hashed if tw["compared_bytes"] else 0),
                   "lz77_decode_kernel": ub + pipe.arc_bytes, "sha1_extents_kernel": ub, "sha1_chain_kernel": ub}
            nhashed = pipe.nfiles - tw["twins"]       # files SHA-256 actually sees: one wave (chain) or one lane each
            waves = {"lz77_decode_kernel": pipe.nb, "sha1_chain_kernel": pipe.nb, "sha256_chain_kernel": nhashed, "sha256_extents_kernel": -(-nhashed // 64),
                     "sha256_group_kernel": -(-nhashed // 4),
                     "lz77_copy_kernel": pipe.nb, "sha1_extents_kernel": -(-pipe.nu // 64)}
            alg_step = pipe.arc_bytes + 2 * pipe.total  # r bytes read + 1 byte written per restored byte + 1 byte read back for SHA-256
            metric = "MB/s compressed archive input extracted + verified (SHA-1 per fragment, SHA-256 per file), Silesia x%d -m1" % a.copies
        else:
            in_bytes = pipe.total * world
            ub = st["unique_bytes"]
            # algorithmic bytes per launch: fragment/hash kernels read every input byte once;
            # the LZ77 and checksum kernels read every unique byte once (+ r bytes written)
            tw = pipe.twin_stats if (pipe.use_twins and pipe.twin_stats) else dict(twins=0, twin_bytes=0, compared=0, compared_bytes=0)
            walked = pipe.total - tw["twin_bytes"]          # bytes the fragment loop and the fragment SHA-1 pass actually see
            alg = {"fragment_spec_kernel": walked, "sha1_extents_kernel": walked,
                   # member bytes streamed once + the representatives they are compared with (cache resident after the first touch)
                   "twin_compare_kernel": tw["compared_bytes"] + (walked if tw["compared_bytes"] else 0), "lz77_spec_kernel": ub // max(1, world) + out_bytes // max(1, world),
                   "lz77_direct_kernel": ub // max(1, world) + out_bytes // max(1, world), "sha1_chain_kernel": ub // max(1, world)}
            waves = {"sha1_chain_kernel": st["blocks"], "lz77_spec_kernel": 3 * -(-ub // (2 << 20)), "lz77_direct_kernel": 4 * st["blocks"]}     # three waves per segment, four per block
            if tw["twin_bytes"]:
                # what is left after the fold is walked by a handful of waves (one lane per 256 KiB segment / per fragment)
                seg_b = max(256 << 10, walked // 158720)
                nfw = max(1, int(st["fragments"] * walked // max(1, pipe.total)))
                waves.update({"fragment_spec_kernel": -(-walked // seg_b // 64) or 1, "fragment_resume_kernel": -(-walked // seg_b // 64) or 1,
                              "sha1_extents_kernel": -(-nfw // 64)})
            alg_step = pipe.total + ub // max(1, world) + out_bytes // max(1, world)   # SURVEY 8(d), per rank: every input byte read once + the unique bytes into the compressor + the output
            metric = ("MB/s compressed output (bit-identical .zpaq) at -m1, Silesia x%d" % a.copies) if a.workload == "silesia_x256_m1" else \
                     ("MB/s compressed output (bit-identical .zpaq) at -m1, %d unique 16 MiB units x%d duplication per GPU" % (a.units, a.dup))
        traffic = {}
        # PMC bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) from the rocprofv3 --pmc passes of this workload (tools/gpu_traffic.sh,
        # profiles/summarize.py): profiles/traffic.json for the headline, traffic_<workload>.json for the others
        tf = os.path.join(ROOT, "profiles", "traffic.json" if a.workload == "silesia_x256_m1" else "traffic_%s.json" % a.workload)
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get("bytes_per_launch", {})
        traffic_key = {"sha1_extents_kernel": "sha1_extents_staged_kernel"}     # (names the kernel trace gives to what the event scopes call ...)

        def roof(k, src=None, nsteps=None, how=None):
            src = kern if src is None else src
            nsteps = prof_steps if nsteps is None else nsteps
            cnt, ms = src[k]
            per = ms / cnt
            ab = alg.get(k)          # algorithmic bytes of the kernel per STEP (a step may launch it more than once)
            if not ab:
                return None
            ach = ab / 1e9 / (ms / nsteps / 1e3)
            tr = traffic.get(k, traffic.get(traffic_key.get(k, k)))
            r = {"bound": "hbm", "kernel": k, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": tr, "avg_launch_ms": round(per, 4),
                 "launches_per_step": round(cnt / nsteps, 2), "algorithmic_bytes_per_step": int(ab), "ms_per_step": round(ms / nsteps, 3)}
            if tr:
                r["traffic_over_algorithmic"] = round(tr * (cnt / nsteps) / ab, 3)
            if how:
                r["measured"] = how
            if k in VALU_OPS_PER_BYTE:
                ceil = LANE_OPS_PER_S / VALU_OPS_PER_BYTE[k] / 1e9
                r["integer_issue_ceiling_GBps"] = round(ceil, 1)
                r["frac_of_integer_ceiling"] = round(ach / ceil, 4)
            if k in waves:
                r["waves"] = int(waves[k])
                r["note"] = "serial chain(s): %d waves of 1024 SIMDs; one instruction per ~4 cycles per wave" % waves[k]
            return r
        # `roofline` = the CHIP-FILLING kernel with the most time per step, from the durations of a job that ran ALONE (HIP
        # events on the kernel's own stream, `single_job`): a kernel time that fits inside ms_per_step, not a latency stretched
        # by eleven other jobs.  `roofline_in_flight` = the same kernel's events in the timed region (what contention does to
        # it); `roofline_longest_chain` = the few-wave kernel with the most time alone (block checksum chains: what bounds ONE
        # job's latency, not the chip's throughput).  With one step in flight (dup8) the timed region IS the job alone.
        simds = 1024
        base, nbase = (kern_alone, n_alone) if kern_alone else (kern, prof_steps)
        how = ("one job alone, %d runs (single_job)" % n_alone) if kern_alone else "the timed region (one step in flight)"
        fill = [k for k in base if alg.get(k) and waves.get(k, simds) >= simds]
        chains = [k for k in base if alg.get(k) and waves.get(k, simds) < simds]
        dom_fill = max(fill, key=lambda k: base[k][1], default=None)
        roof_dom = roof(dom_fill, base, nbase, how) if dom_fill else None
        roof_fill = roof(dom_fill) if (dom_fill and kern_alone and dom_fill in kern) else None
        roof_chain = roof(max(chains, key=lambda k: base[k][1]), base, nbase, how) if chains else None
        roof_all = [r for r in (roof(k, base, nbase, how) for k in sorted(base, key=lambda k: -base[k][1])) if r] if a.roofline_all else None
        e2e = alg_step / 1e9 / sec
        res = {"metric": metric, "value": round(out_bytes / 1e6 / sec, 3),
               "unit": "MB/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": round(sec * 1e3, 3),
               "higher_is_better": True, "scaling": "strong" if shared else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": a.workload if a.workload != "silesia_x256_m1" else "silesia_x%d_m1" % a.copies,
                          **({"corpus": "one Silesia x%d split by file range over %d ranks" % (a.copies, world)} if shared else {}),
                          "files": pipe.nfiles * world, "input_bytes": in_bytes,
                          "method": "14 -> x4,1,5,0,3,24", "block_bytes": BLOCK_LIMIT, "fragment": 6, **st},
               "timed_step": ("one C-ABI call per rank: zpqj_add_sharded_dev (this rank's files resident in HBM in, the whole job's archive out on every rank) with "
                              "the in-tree RCCL collectives inside it (zpqr_allgatherv / zpqr_allgatherv_dev as C function pointers, one communicator per add in flight: "
                              "no torch collective and no Python in the timed region's data path)"
                              if sharded_product and a.dist_backend != "gloo" else
                              "one C-ABI call per rank: zpqj_add_sharded_dev over functional stand-in collectives (gloo: test only)" if sharded_product else
                              "one C-ABI call: zpqj_extract_dev (whole journaling archive resident in HBM in -- the call reads the index itself --, restored "
                              "files and their SHA-256 left in HBM) + the digest compare" if (extract and getattr(pipe, "product", False)) else
                              "one C-ABI call: zpqj_add_dev (files resident in HBM in, whole journaling archive -- c, d, h, i blocks -- out to host memory)"
                              if product else "call-by-call orchestration in bench.py (d blocks only)"),
               "identity": "per d block and per table: every block, fragment boundary, SHA-1 and the dedup map equal the reference-derived oracle; "
                           "whole-archive identity is not provable here (block cut rule, R,t hint and file order of the missing zpaqfranz.cpp are unpinned)",
               ("output_GBps" if extract else "input_GBps"): round(in_bytes / 1e9 / sec, 3), "steps_in_flight": depth,
               "ms_per_step_serial": round(stagger[0] * 1e3, 3) if depth > 1 else round(sec * 1e3, 3),      # = single_job.ms
               **({"ms_per_step_cold": round(cold * 1e3, 3), "value_cold": round(out_bytes / 1e6 / cold, 3), "timing": STEADY_NOTE,
                   "jobs_between_barriers": prof_steps,
                   "completions_s": completions if len(completions) <= 64 else completions[:: max(1, len(completions) // 64)]} if cold else {}),
               "kernels_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
               **({"product_one_call": product_once} if product_once else {}),
               "roofline": roof_dom, "roofline_in_flight": roof_fill, "roofline_longest_chain": roof_chain, "single_job": single,
               **({"kernels_ms_per_job_alone": {k: round(v[1] / n_alone, 3) for k, v in sorted(kern_alone.items(), key=lambda kv: -kv[1][1])}} if kern_alone else {}),
               **({"roofline_all": roof_all} if roof_all else {}),
               "roofline_end_to_end": {"bound": "hbm", "achieved": round(e2e, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(e2e / HBM_PEAK_GBS, 5),
                                       "algorithmic_bytes_per_step": int(alg_step),
                                       "note": "whole step: SURVEY 8(d) algorithmic bytes (input once + unique + output; extract: r + 1 + 1) / ms_per_step; the passes are integer-issue bound, not HBM bound: see integer_issue_ceiling_GBps per kernel"}}
        if extract:
            res["sha256_mismatches"] = pipe.sha256_mismatches
            res["archive_bytes"] = {"whole_archive": int(out_bytes), "d_blocks": int(pipe.arc_bytes)}
            res["value_d_blocks_only"] = round(pipe.arc_bytes / 1e6 / sec, 3)      # (what rounds 2-5 quoted: the d blocks' bytes per second)
            if world == 1 and not a.no_verify and not os.environ.get("ZPQ_BENCH_NO_VARIANT"):
                # the same job with the fold the other way round (default: fold ON beside the fold-off headline)
                main_fold = bool(getattr(pipe, "use_twins", False))
                for r_ in runners:
                    r_.use_twins = not main_fold
                try:
                    n2 = max(len(runners), min(steps, 8))
                    dt2, _, ob2, _ = steady_window(run_steps, barrier, depth if len(runners) > 1 else 1, 0, n2)
                    sec2 = dt2 / n2
                    tws = runners[0].twin_stats or {}
                    res["every_byte_hashed" if main_fold else "twin_fold_on"] = dict(
                        ms_per_step=round(sec2 * 1e3, 3), value=round(ob2 / 1e6 / sec2, 3), unit="MB/s", steps=n2, output_GBps=round(pipe.total / 1e9 / sec2, 3),
                        sha256_mismatches=int(sum(r_.sha256_mismatches for r_ in runners)), **{k: tws[k] for k in ("twins", "twin_bytes") if k in tws})
                finally:
                    for r_ in runners:
                        r_.use_twins = main_fold
                    pipe.step()          # (what the verification below reads -- digests, restored bytes -- is the headline mode's again)
                    if not main_fold:
                        pipe.twin_stats = None
            res["twin_fold"] = dict(enabled=bool(getattr(pipe, "use_twins", True)), **tw,
                                    note="restored files whose bytes equal an earlier restored file's (every byte compared on the device) take that "
                                         "file's SHA-256; the representatives are hashed (zpq_sha256_files_dev)")
            if world == 1 and not a.no_verify:
                # the same extract with BLAKE3 as the per-file check (what zpaqfranz offers beside SHA-256 / XXHASH64)
                import orc
                want12 = [orc.blake3(b) for _, b in corpus]
                for r_ in runners:
                    r_.verify_hash = "blake3"; r_.blake3_want = want12 * a.copies
                nb3 = 2 * len(runners)
                dtb, _, _, _ = steady_window(run_steps, barrier, depth if len(runners) > 1 else 1, 0, nb3)
                secb = dtb / nb3
                res["blake3_verify"] = {"ms_per_step": round(secb * 1e3, 3), "value": round(pipe.arc_bytes / 1e6 / secb, 3), "unit": "MB/s",
                                        "output_GBps": round(pipe.total / 1e9 / secb, 3),
                                        "mismatches": int(sum(getattr(r_, "blake3_mismatches", 0) for r_ in runners)),
                                        "note": "same job, BLAKE3 (tree hash, checked against the oracle's digests of the 12 members) instead of SHA-256 per file"}
                for r_ in runners:
                    r_.verify_hash = "sha256"
        threads = min(32, len(os.sched_getaffinity(0)))
        if world == 1 and not a.no_verify and not a.force_collectives:
            if extract:
                # every restored file against hashlib over the originals (the device compare above used the same expectations;
                # here the restored BYTES of the first copy are pulled back and hashed on the host as well)
                import hashlib
                got = bytes(pipe.d_sha_got[: pipe.nfiles * 32].cpu().numpy())
                want = b"".join(hashlib.sha256(b).digest() for _, b in corpus) * a.copies
                unit = layout["unit"]
                host_copy = bytes(pipe.out[:unit].cpu().numpy())
                res["verified_all_files"] = bool(got == want and pipe.sha256_mismatches == 0 and host_copy == b"".join(b for _, b in corpus))
                res["verified_all_blocks"] = True           # every d block decoded with its stored SHA-1 matching (step() raises otherwise)
            else:
                res.update(verify_add(pipe, layout, corpus, threads))
                if product_archive is not None:
                    res.update(verify_product(pipe, product_archive, threads))
        product_archive = None          # (a view of memory the next product step releases)
        if not extract:
            res["twin_fold"] = dict(enabled=bool(pipe.use_twins), **(pipe.twin_stats or {}),
                                    note="files whose bytes equal an earlier file's are found by comparing every byte on the device (HBM-bound) "
                                         "before anything is hashed; only the representatives go through the fragment loop and SHA-1, twins take "
                                         "their records: identical tables for any input (zpq_fragment_sha1_dev, csrc/twins.hip)")
        if (a.workload == "silesia_x256_m1" and world == 1 and not a.force_collectives and not a.dump_archive
                and not (os.environ.get("ZPQ_BENCH_NO_VARIANT") or os.environ.get("ZPQ_BENCH_NO_PLAIN"))):
            # the same job with the twin fold the other way round: by default that is fold ON (files equal to an earlier file are
            # found by comparing every byte and take its records: the corpus-shaped optimisation, reported beside the headline)
            main_fold = pipes[0].use_twins
            for p_ in pipes:
                p_.use_twins = not main_fold
            try:
                n2 = max(len(pipes), min(steps, 36))
                dt2, _, ob2, _ = steady_window(run_steps, barrier, depth if len(pipes) > 1 else 1, 0, n2)
                sec2 = dt2 / n2
                var = {"ms_per_step": round(sec2 * 1e3, 3), "value": round(ob2 / 1e6 / sec2, 3), "unit": "MB/s", "steps": n2,
                       "input_GBps": round(in_bytes / 1e9 / sec2, 3)}
                if main_fold:
                    res["every_byte_hashed"] = dict(var, note="twin fold off: every file fragmented and SHA-1'd; same tables, same blocks")
                else:
                    tws = pipes[0].twin_stats or {}
                    res["twin_fold_on"] = dict(var, **{k: tws[k] for k in ("twins", "twin_bytes") if k in tws},
                                               note="same job with the twin fold (--twins): files whose bytes equal an earlier file's are found by "
                                                    "comparing every byte on the device and take its fragment records; same tables, same blocks; "
                                                    "corpus-shaped (x256 replication), hence not the headline")
                    res["value_twin_fold"] = var["value"]
            finally:
                for p_ in pipes:
                    p_.use_twins = main_fold
        if world == 1 and not a.no_cpu_baseline:
            base = b"".join(b for _, b in corpus)
            if a.workload == "silesia_x256_m1":
                res["cpu_baseline"] = cpu_baseline("add", [a.copies, json.dumps(layout["sizes"])], [base])
            elif a.workload == "dup8_m1":
                sample = max(8, min(64, a.units))
                res["cpu_baseline"] = cpu_baseline("dup8", [sample, a.dup, a.units], [layout["pool_bytes"]])
            else:
                # index for the CPU run: blocks, unique fragments, the 12 members' pointer lists
                ex = pipe
                nm = len(corpus)
                upos = ex.d_upos.cpu().numpy(); ulen = ex.d_ulen.cpu().numpy()
                ublk = np.searchsorted(np.array(ex.plain_off + [1 << 62]), upos, side="right") - 1
                src = ex.d_src.cpu().numpy()
                uid = {int(p): i for i, p in enumerate(upos.tolist())}
                fo = np.array(ex.file_off); dst = ex.d_dst.cpu().numpy()
                fidx = np.searchsorted(fo, dst, side="right") - 1
                members = [[uid[int(s)] for s in src[fidx == m].tolist()] for m in range(nm)]
                blocks_blob = b"".join(bytes(ex.arc[o:o + n].cpu().numpy()) for o, n in zip(ex.arc_off, ex.blk_len))
                boff = np.concatenate(([0], np.cumsum(ex.blk_len))).tolist()
                index = dict(block_off=boff, block_usize=ex.usize, frag_block=ublk.tolist(),
                             frag_off=(upos - np.array(ex.plain_off)[ublk]).tolist(), frag_len=ulen.tolist(), members=members, copies=a.copies)
                res["cpu_baseline"] = cpu_baseline("extract", [], [blocks_blob, json.dumps(index).encode()])
        print(json.dumps(res))
    if a.dump_archive and rank == 0 and a.workload != "extract_m1":
        if sharded_product:
            # the d blocks of the archive every rank got back (between its c block and its first h block)
            blks = split_archive(pipe.archive, payloads="")
            parts = [bytes(pipe.archive[b_[2]:b_[3]]) for b_ in blks if b_[0][17:18] == b"d"]
        elif world > 1:
            parts = [bytes(g.cpu().numpy()) for g in pipe.gathered]
        else:
            parts = [b for _, b in pipe.framed_blocks()]
        # blocks are owned in ascending order by ascending rank: concatenating the per-rank streams IS block order
        with open(a.dump_archive, "wb") as f:
            for g in parts:
                f.write(g)
    if dist.is_initialized():
        dist.destroy_process_group()
    for p_ in pipes:
        p_.drop_archive()
    for g_ in gathers:
        g_.close()
    for e_ in engines:
        e_.close()


if __name__ == "__main__":
    main()
