"""world_size-2 gloo test of the multi-GPU plan: both ranks all-gather their fragment
tables, resolve the global first-occurrence dedup identically, and derive matching ownership and
peer-to-peer lists.  Pure CPU."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zpaqfranz_amd import sharding


def _tables(rank):
    """Fragment table of one rank: 2500 fragments, ~60 % of the digests shared between ranks."""
    rng = np.random.default_rng(100 + rank)
    shared = np.random.default_rng(7).integers(0, 256, size=(1500, 20), dtype=np.uint8)
    own = rng.integers(0, 256, size=(1000, 20), dtype=np.uint8)
    pick = rng.integers(0, 2500, size=2500)
    dig = np.where((pick < 1500)[:, None], shared[np.minimum(pick, 1499)], own[np.minimum(np.maximum(pick - 1500, 0), 999)])
    lens = (np.frombuffer(dig.tobytes(), dtype=np.uint8).reshape(-1, 20)[:, 0].astype(np.int64) * 2000 + 4096)
    return dig, lens


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dig, lens = _tables(rank)
    td, tl = torch.from_numpy(dig.copy()), torch.from_numpy(lens.copy())
    gd = [torch.empty_like(td) for _ in range(world)]
    gl = [torch.empty_like(tl) for _ in range(world)]
    dist.all_gather(gd, td)
    dist.all_gather(gl, tl)
    all_d = torch.cat(gd).numpy()
    all_l = torch.cat(gl).numpy()
    first = sharding.first_occurrence(all_d)
    p = sharding.plan(first, all_l, [len(dig)] * world, rank, block_limit=4 << 20)
    summary = dict(rank=rank, nblocks=p["nblocks"], mine=p["mine"].tolist(),
                   send={k: v.tolist() for k, v in p["send"].items()}, recv={k: v.tolist() for k, v in p["recv"].items()},
                   uniq=int(len(p["uniq_idx"])), blk_sig=int(p["blk"].sum()) if len(p["blk"]) else 0)
    out = [None] * world
    dist.all_gather_object(out, summary)
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


def test_two_rank_plan_is_consistent():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    assert a["nblocks"] == b["nblocks"] and a["uniq"] == b["uniq"] and a["blk_sig"] == b["blk_sig"]
    assert sorted(a["mine"] + b["mine"]) == list(range(a["nblocks"]))          # every block has exactly one owner
    assert a["send"].get(1, []) == b["recv"].get(0, [])                       # what 0 sends is what 1 expects
    assert b["send"].get(0, []) == a["recv"].get(1, [])
    # and the plan equals the single-process plan over the concatenated tables
    d0, l0 = _tables(0); d1, l1 = _tables(1)
    first = sharding.first_occurrence(np.concatenate((d0, d1)))
    solo = sharding.plan(first, np.concatenate((l0, l1)), [len(d0) + len(d1)], 0, block_limit=4 << 20)
    assert solo["nblocks"] == a["nblocks"] and len(solo["uniq_idx"]) == a["uniq"]


def test_pack_blocks_respects_limit():
    rng = np.random.default_rng(1)
    lens = rng.integers(4096, 520193, size=5000)
    blk, nb = sharding.pack_blocks(lens)
    assert nb == blk.max() + 1 and (np.diff(blk) >= 0).all()
    for b in range(nb):
        m = blk == b
        assert lens[m].sum() + 4 * m.sum() + 8 <= sharding.BLOCK_LIMIT
    # greedy: adding the next fragment to any block but the last would overflow
    for b in range(nb - 1):
        m = blk == b
        nxt = lens[np.nonzero(blk == b + 1)[0][0]]
        assert lens[m].sum() + 4 * m.sum() + 8 + nxt + 4 > sharding.BLOCK_LIMIT


def test_balanced_ownership_deals_the_blocks_out_and_the_lists_match():
    """ONE corpus split by file range (bench.py's default with several ranks): every new fragment sits on rank 0.  Ownership
    by residence gives rank 0 every block; balance=True deals contiguous block ranges to the ranks, and what a rank sends is
    what the owner expects, fragment for fragment."""
    rng = np.random.default_rng(3)
    n0 = 3000
    dig0 = rng.integers(0, 256, size=(n0, 20), dtype=np.uint8)
    lens0 = rng.integers(4096, 200000, size=n0).astype(np.int64)
    world = 4
    dig = np.concatenate([dig0] * world); lens = np.concatenate([lens0] * world)          # ranks 1.. hold copies of rank 0's files
    first = sharding.first_occurrence(dig)
    plans = [sharding.plan(first, lens, [n0] * world, r, block_limit=4 << 20) for r in range(world)]
    assert len(plans[0]["mine"]) == plans[0]["nblocks"] and all(len(p["mine"]) == 0 for p in plans[1:])
    bal = [sharding.plan(first, lens, [n0] * world, r, block_limit=4 << 20, balance=True, local_copies=False) for r in range(world)]
    nb = bal[0]["nblocks"]
    assert nb >= 2 * world
    owned = [p["mine"].tolist() for p in bal]
    assert sorted(sum(owned, [])) == list(range(nb))
    assert all(abs(len(o) - nb / world) <= 1 for o in owned)
    assert all(owned[r] == sorted(owned[r]) and (not owned[r] or not owned[r + 1] or owned[r][-1] < owned[r + 1][0]) for r in range(world - 1))
    for src in range(world):
        for dst in range(world):
            if src != dst:
                a = bal[src]["send"].get(dst, np.zeros(0, dtype=np.int64)); b = bal[dst]["recv"].get(src, np.zeros(0, dtype=np.int64))
                assert a.tolist() == b.tolist()
    assert sum(len(v) for v in bal[0]["send"].values()) > 0 and not bal[1]["send"]       # rank 0 ships the blocks it does not keep


def test_a_rank_reads_the_fragments_of_its_blocks_from_its_own_copy_where_it_has_one():
    """local_copies (the default with balance=True): whole copies on every rank -> nothing is shipped, every fragment of a dealt
    block is read at the rank's own occurrence; a rank that lacks some fragments (its copy is cut short) gets exactly those, and
    sender and receiver agree on the lists.  The block contents (which unique fragments, in which order) never change."""
    rng = np.random.default_rng(4)
    n0 = 3000
    dig0 = rng.integers(0, 256, size=(n0, 20), dtype=np.uint8)
    lens0 = rng.integers(4096, 200000, size=n0).astype(np.int64)
    world = 4
    dig = np.concatenate([dig0] * world); lens = np.concatenate([lens0] * world)
    first = sharding.first_occurrence(dig)
    ref = [sharding.plan(first, lens, [n0] * world, r, block_limit=4 << 20, balance=True, local_copies=False) for r in range(world)]
    loc = [sharding.plan(first, lens, [n0] * world, r, block_limit=4 << 20, balance=True) for r in range(world)]
    for r in range(world):
        assert not loc[r]["send"] and not loc[r]["recv"]
        assert loc[r]["mine"].tolist() == ref[r]["mine"].tolist()
        for b in loc[r]["mine"].tolist():
            u, l, sr, si = loc[r]["blocks"][b]
            assert u.tolist() == ref[r]["blocks"][b][0].tolist() and (sr == r).all()
            assert (si >= r * n0).all() and (si < (r + 1) * n0).all() and np.array_equal(first[si], u) and np.array_equal(lens[si], l)
    # rank 2 holds only the first 40 % of a copy, rank 3 a copy in another order
    k = int(0.4 * n0)
    perm = rng.permutation(n0)
    dig2 = np.concatenate([dig0, dig0, dig0[:k], dig0[perm]]); lens2 = np.concatenate([lens0, lens0, lens0[:k], lens0[perm]])
    counts = [n0, n0, k, n0]
    first2 = sharding.first_occurrence(dig2)
    pl = [sharding.plan(first2, lens2, counts, r, block_limit=4 << 20, balance=True) for r in range(world)]
    lo = np.concatenate(([0], np.cumsum(counts)))
    for src in range(world):
        for dst in range(world):
            if src != dst:
                a = pl[src]["send"].get(dst, np.zeros(0, dtype=np.int64)); b = pl[dst]["recv"].get(src, np.zeros(0, dtype=np.int64))
                assert a.tolist() == b.tolist()
    assert not pl[1]["recv"] and not pl[3]["recv"] and set(pl[2]["recv"]) == {0}           # only the short copy needs anything
    got = pl[2]["recv"][0]
    assert (got >= k).all() and len(got) == sum(int((pl[2]["blocks"][b][2] != 2).sum()) for b in pl[2]["mine"].tolist())
    for r in range(world):
        for b in pl[r]["mine"].tolist():
            u, l, sr, si = pl[r]["blocks"][b]
            own = sr == r
            assert (si[own] >= lo[r]).all() and (si[own] < lo[r + 1]).all() and np.array_equal(first2[si], u)
