"""zpqj_add_sharded: the journaling add across PROCESSES, one GPU each, with ONE caller-supplied
collective (an all-gather of byte strings).  CPU part: the file plan and the torch.distributed all-gather helper
(gloo, world 2).  GPU part: two processes on the same device produce, each, exactly the archive one GPU writes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _corpus(seed=11, nfiles=40):
    """Files of 0..700 KB with repeats inside and across files (so dedup pointers cross the range edge) and one file
    made of pieces of the others."""
    rng = np.random.default_rng(seed)
    pool = [rng.integers(0, 64, size=int(rng.integers(20000, 300000)), dtype=np.uint8).tobytes() for _ in range(12)]
    files = []
    for i in range(nfiles):
        k = int(rng.integers(0, 4))
        body = b"".join(pool[int(rng.integers(0, 12))] for _ in range(k))
        if i % 7 == 3:
            body += rng.integers(0, 256, size=int(rng.integers(1, 90000)), dtype=np.uint8).tobytes()
        files.append(("dir%d/f%03d.bin" % (i % 3, (i * 17) % 101), body))
    return files


def test_shard_files_is_a_partition_into_contiguous_name_ranges():
    from zpaqfranz_amd import build, engine
    build.build(verbose=False)
    files = _corpus()
    names, sizes = [f[0] for f in files], [len(f[1]) for f in files]
    order = sorted(range(len(files)), key=lambda i: names[i].encode())
    for world in (1, 2, 3, 8, 64):
        marks = [engine.jidac_shard_files(names, sizes, world, r) for r in range(world)]
        owner = [[r for r in range(world) if marks[r][i]] for i in range(len(files))]
        assert all(len(o) == 1 for o in owner)                                  # every file exactly once
        seq = [owner[i][0] for i in order]
        assert seq == sorted(seq)                                               # contiguous ranges in name order
        if world == 2:
            half = sum(sizes[i] for i in order if owner[i][0] == 0)
            assert abs(half - sum(sizes) / 2) <= max(sizes)
    with pytest.raises(engine.ZpqError):
        engine.jidac_shard_files(names, sizes, 2, 2)


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zpaqfranz_amd import engine
    ag = engine.dist_allgather_bytes()
    got = [ag(b"" if rank else b"x" * 70001), ag(bytes([rank]) * (3 + rank)), ag(b"")]
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_of_byte_strings_over_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    want = [[b"x" * 70001, b""], [b"\0" * 3, b"\1" * 4], [b"", b""]]
    assert res[0] == want and res[1] == want


def _add_worker(rank, world, port, q, method, flags):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zpaqfranz_amd import engine
    files = _corpus()
    names, sizes = [f[0] for f in files], [len(f[1]) for f in files]
    mine = engine.jidac_shard_files(names, sizes, world, rank)
    part = [(n, b if m else None, len(b)) for (n, b), m in zip(files, mine)]      # other ranks' data is never passed in
    eng = engine.Engine(0)
    arc, st = engine.jidac_add_sharded(eng, rank, world, engine.dist_allgather_bytes(), None, part, 20260925120000, method, **flags)
    # a second version on top: half of the files changed, the old fragments are known from the archive
    files2 = [(n, (b[:len(b) // 2] + b"new" + b[len(b) // 2:]) if i % 2 else b) for i, (n, b) in enumerate(files)]
    sizes2 = [len(f[1]) for f in files2]
    mine2 = engine.jidac_shard_files(names, sizes2, world, rank)
    part2 = [(n, b if m else None, len(b)) for (n, b), m in zip(files2, mine2)]
    arc2, st2 = engine.jidac_add_sharded(eng, rank, world, engine.dist_allgather_bytes(), arc, part2, 20260925130000, method, **flags)
    q.put((rank, arc, st, arc2, st2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("method,flags", [("14", {}), ("1", {"checksums": True}), ("24", {"hint": True})])
def test_two_processes_write_the_single_gpu_archive(method, flags):
    from zpaqfranz_amd import engine
    files = _corpus()
    eng = engine.Engine(0)
    want, wst = engine.jidac_add(eng, None, files, 20260925120000, method, **flags)
    files2 = [(n, (b[:len(b) // 2] + b"new" + b[len(b) // 2:]) if i % 2 else b) for i, (n, b) in enumerate(files)]
    want2, wst2 = engine.jidac_add(eng, want, files2, 20260925130000, method, **flags)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_add_worker, args=(r, 2, port, q, method, flags)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
    for r in range(2):
        arc, st, arc2, st2 = res[r]
        assert arc == want and st == wst, r
        assert arc2 == want2 and st2 == wst2, r
    assert wst["d_blocks"] >= 1 and wst2["new_fragments"] > 0
    got = engine.jidac_extract(eng, want + want2)
    assert got == dict(files2)


@pytest.mark.gpu
def test_a_failing_collective_fails_the_add():
    from zpaqfranz_amd import engine
    files = _corpus(nfiles=6)
    part = [(n, b, len(b)) for n, b in files]
    eng = engine.Engine(0)

    def broken(_):
        raise RuntimeError("link down")
    with pytest.raises(RuntimeError, match="link down"):
        engine.jidac_add_sharded(eng, 0, 1, broken, None, part, 20260925120000, "14")
    arc, _ = engine.jidac_add_sharded(eng, 0, 1, lambda b: [b], None, part, 20260925120000, "14")     # world 1: the plain add
    assert arc == engine.jidac_add(eng, None, files, 20260925120000, "14")[0]


def test_rccl_gather_library_exports_every_symbol_of_its_header():
    """zpaqfranz_amd/shim/rccl_gather.h (the in-tree collective: rccl.h, no torch) vs libzpaq_rccl.so -- loads without a GPU."""
    import ctypes
    import re
    from zpaqfranz_amd import build
    build.build(verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(build.RCCL_SO):
        build.build_rccl()
    hdr = open(os.path.join(root, "zpaqfranz_amd", "shim", "rccl_gather.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(zpqr_\w+)\s*\(", hdr))
    assert {"zpqr_unique_id", "zpqr_create", "zpqr_allgatherv", "zpqr_last_error", "zpqr_destroy"} <= names
    R = ctypes.CDLL(build.RCCL_SO)
    for n in sorted(names):
        assert hasattr(R, n), n
    # its signature is the one zpqj_add_sharded takes
    jh = open(os.path.join(root, "zpaqfranz_amd", "shim", "jidac_gpu.h")).read()
    assert "typedef int (*zpqj_allgatherv_fn)(void* user, const void* send, size_t send_len, void** recv, size_t* recv_len);" in jh
    assert "int zpqr_allgatherv(void* comm, const void* send, size_t send_len, void** recv, size_t* recv_len);" in hdr


@pytest.mark.gpu
def test_sharded_add_over_the_in_tree_rccl_collective_world_size_1():
    """The C function zpqr_allgatherv (RCCL on the engine's stream) handed to zpqj_add_sharded as its collective: with one rank
    the archive is the plain add's; the collective itself returns what was sent (strings of 0, 1 and 70 001 bytes)."""
    if os.environ.get("ZPQ_TEST_EMU") == "1":
        pytest.skip("RCCL needs the device")
    from zpaqfranz_amd import engine
    files = _corpus(nfiles=9)
    part = [(n, b, len(b)) for n, b in files]
    eng = engine.Engine(0)
    g = engine.RcclGather(eng, 0, 1, engine.RcclGather.unique_id())
    try:
        for s in (b"", b"x", bytes(range(256)) * 273 + b"!"):
            assert g(s) == [s]
        arc, st = engine.jidac_add_sharded(eng, 0, 1, g, None, part, 20260925120000, "14", checksums=True)
        want, wst = engine.jidac_add(eng, None, files, 20260925120000, "14", checksums=True)
        assert arc == want and st == wst
    finally:
        g.close()
        eng.close()


def _add_dev_worker(rank, world, port, q, method, flags, with_dev):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zpaqfranz_amd import engine
    files = sorted(_corpus(), key=lambda f: f[0].encode())
    names, sizes = [f[0] for f in files], [len(f[1]) for f in files]
    mine = engine.jidac_shard_files(names, sizes, world, rank)
    eng = engine.Engine(0)
    buf = eng.upload(b"".join(b for (n, b), m in zip(files, mine) if m))       # this rank's range, back to back in name order
    sections = []

    def wrap(k, thunk):                  # (what bench.py orders the collectives of its adds in flight with)
        sections.append(k)
        return thunk()
    arc, st = engine.jidac_add_sharded_dev(eng, rank, world, engine.dist_allgather_bytes(), engine.dist_allgather_dev(eng) if with_dev else None,
                                           None, names, sizes, buf.ptr, 20260925120000, method, wrap=wrap, **flags)
    q.put((rank, arc, st, sections))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("with_dev", [True, False])
def test_two_processes_with_files_resident_in_hbm_write_the_single_gpu_archive(with_dev):
    """zpqj_add_sharded_dev: every rank's files already in HBM; with_dev: the compressed d blocks travel through the DEVICE form of
    the collective (HBM pointers in, HBM pointers out) and only their sizes through the host form."""
    from zpaqfranz_amd import engine
    files = sorted(_corpus(), key=lambda f: f[0].encode())
    eng = engine.Engine(0)
    flags = {"checksums": True}
    want, wst = engine.jidac_add(eng, None, files, 20260925120000, "14", **flags)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_add_dev_worker, args=(r, 2, port, q, "14", flags, with_dev)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
    for r in range(2):
        arc, st, sections = res[r]
        assert arc == want and st == wst, r
        assert sections == list(range(4 if with_dev else 3)), sections         # tables, seam fragments, sizes + blocks | blocks
    assert engine.jidac_extract(eng, want) == dict(files)


@pytest.mark.gpu
def test_sharded_dev_add_over_the_in_tree_rccl_collectives_world_size_1():
    """zpqj_add_sharded_dev with zpqr_allgatherv + zpqr_allgatherv_dev (RCCL, HBM -> HBM) as its collectives, one rank: the plain
    add's archive; the device collective returns what was sent."""
    if os.environ.get("ZPQ_TEST_EMU") == "1":
        pytest.skip("RCCL needs the device")
    from zpaqfranz_amd import engine
    files = sorted(_corpus(nfiles=9), key=lambda f: f[0].encode())
    names, sizes = [f[0] for f in files], [len(f[1]) for f in files]
    eng = engine.Engine(0)
    g = engine.RcclGather(eng, 0, 1, engine.RcclGather.unique_id())
    buf = eng.upload(b"".join(b for _, b in files))
    try:
        payload = bytes(range(256)) * 999 + b"tail"
        d = eng.upload(payload)
        (ptr, n), = g.gather_dev(d.ptr, len(payload))
        import ctypes as C
        out = C.create_string_buffer(n)
        assert eng.L.zpq_d2h(eng.ctx, out, C.c_void_p(ptr), n) == 0 and out.raw == payload
        assert g.gather_dev(0, 0) == [(0, 0)]
        want, wst = engine.jidac_add(eng, None, files, 20260925120000, "14", checksums=True)
        for wrap in (None, lambda k, thunk: thunk()):            # the C functions handed over directly / through the ordering hook
            arc, st = engine.jidac_add_sharded_dev(eng, 0, 1, g, None, None, names, sizes, buf.ptr, 20260925120000, "14", checksums=True, wrap=wrap)
            assert arc == want and st == wst
    finally:
        g.close()
        eng.close()


def _tree_of_copies():
    """two directories with the same files (every fragment of the second is a duplicate of one in the first) + a few files of
    its own in each: the d blocks are dealt out, the context dealt to reads its own copies"""
    base = sorted(_corpus(seed=23, nfiles=30), key=lambda f: f[0].encode())
    extra = _corpus(seed=29, nfiles=4)
    # (b/ has two files of its own at its END: the last block then starts on rank 1 and mixes fragments of both ranks)
    files = [("a/" + n, b) for n, b in base]
    files += [("b/" + n, b) for n, b in base] + [("b/zz_own_%d" % i, b[:150000]) for i, (_, b) in enumerate(extra[2:])]
    return sorted(files, key=lambda f: f[0].encode())


def _copies_worker(rank, world, port, q, with_dev):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zpaqfranz_amd import engine
    files = _tree_of_copies()
    names, sizes = [f[0] for f in files], [len(f[1]) for f in files]
    mine = engine.jidac_shard_files(names, sizes, world, rank)
    eng = engine.Engine(0)
    buf = eng.upload(b"".join(b for (n, b), m in zip(files, mine) if m))
    arc, st = engine.jidac_add_sharded_dev(eng, rank, world, engine.dist_allgather_bytes(), engine.dist_allgather_dev(eng) if with_dev else None,
                                           None, names, sizes, buf.ptr, 20260925120000, "10")
    q.put((rank, arc, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("with_dev", [True, False])
def test_blocks_of_a_tree_of_copies_are_dealt_out_and_the_archive_stays_the_same(with_dev):
    """One tree of copies over two ranks (the BASELINE metric's shape): every new fragment first occurs on rank 0, the d blocks
    (method "10": 1 MiB blocks, several of them) are dealt out and rank 1 compresses its share from its OWN copies; the archive is
    the single-GPU archive, and so is what several contexts in one process write (zpqj_add_multi)."""
    from zpaqfranz_amd import engine
    files = _tree_of_copies()
    eng = engine.Engine(0)
    want, wst = engine.jidac_add(eng, None, files, 20260925120000, "10")
    assert wst["d_blocks"] >= 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_copies_worker, args=(r, 2, port, q, with_dev)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
    for r in range(2):
        assert res[r][0] == want and res[r][1] == wst, r
    if with_dev:
        e2, e3 = engine.Engine(0), engine.Engine(0)
        try:
            for engs in ([eng, e2], [eng, e2, e3]):
                got, gst = engine.jidac_add(engs, None, files, 20260925120000, "10")
                assert got == want and gst == wst, len(engs)
        finally:
            e2.close(); e3.close()
    assert engine.jidac_extract(eng, want) == dict(files)
