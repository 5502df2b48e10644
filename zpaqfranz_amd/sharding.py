"""Host-side planning of the multi-GPU add path: pure numpy, no device code, so that
it can be exercised with world_size-2 gloo tests on CPU.

Inputs are the all-gathered fragment tables (rank-major, order preserving): `first[i]` = index of the
first fragment with the same SHA-1 (from zpq_dedup_dev, or any exact first-occurrence dedup), `lens[i]`,
and `counts[r]` = number of fragments rank r contributed.  Every rank runs the same deterministic plan."""
import numpy as np

BLOCK_LIMIT = (1 << 24) - 4096   # zpaqfranz -m1: method "14" -> 2^24 - 4096 byte blocks


def pack_blocks(uniq_len, block_limit=BLOCK_LIMIT):
    """Deterministic block packer over the unique-fragment sequence (host logic, as in the reference's
    Jidac::add): a block takes fragments while bytes + 4*count + 8 <= block_limit (the d block carries a
    4-byte size per fragment plus 8 trailer bytes, ZSFX/zsfx.cpp:1468-1500 reads them back).  Returns
    (block id of every unique fragment, number of blocks).  The exact cut rule lives in the missing
    zpaqfranz.cpp: parity unpinned."""
    n = len(uniq_len)
    blk = np.empty(n, dtype=np.int64)
    cs = np.concatenate(([0], np.cumsum(np.asarray(uniq_len, dtype=np.int64) + 4)))
    i, b = 0, 0
    while i < n:
        j = int(np.searchsorted(cs, cs[i] + block_limit - 8, side="right")) - 1
        j = max(j, i + 1)
        blk[i:j] = b
        i, b = j, b + 1
    return blk, b


def plan(first, lens, counts, rank, block_limit=BLOCK_LIMIT, balance=False, local_copies=None):
    """Returns a dict describing what `rank` has to do:
      uniq_idx     global indices of new fragments (ascending)
      nblocks      number of d blocks in the archive
      mine         block ids this rank compresses (owner = rank holding the block's first fragment; balance=True: the d
                   blocks are dealt out in equal contiguous ranges instead, rank r taking blocks [r n / N, (r+1) n / N) --
                   for ONE corpus split over the ranks by file range every NEW fragment sits on the first ranks (a copy
                   duplicates an earlier file), and ownership by residence would leave the compressor to rank 0 alone;
                   ascending ranks still own ascending blocks, so the per-rank streams concatenate to block order)
      blocks       {block id: (global fragment indices, lengths, owning rank of each fragment)}
      send         {dst rank: global indices of MY fragments that live in blocks owned by dst, in order}
      recv         {src rank: global indices of fragments I need from src, in order}
    local_copies (default: on with balance): a fragment of a block dealt to rank r that first occurred on another rank is taken
    from r's OWN data when r holds a duplicate of it (the global table says which of r's fragments have that first
    occurrence) -- every rank can tell that for every rank, so sender and receiver drop it from their lists alike.  With ONE
    corpus of whole copies split by file range every rank holds every unique fragment: nothing is shipped at all, where
    rank 0 would otherwise send every block it does not keep.  blocks[b] then carries a fourth array: the global index of
    the occurrence to read each fragment from (its own index where the source is its first occurrence)."""
    first = np.asarray(first)
    lens = np.asarray(lens, dtype=np.int64)
    ntot = len(first)
    is_new = first == np.arange(ntot, dtype=first.dtype)
    uniq_idx = np.nonzero(is_new)[0]
    blk, nblk = pack_blocks(lens[uniq_idx], block_limit)
    bounds = np.cumsum(np.asarray(counts, dtype=np.int64))
    owner = np.searchsorted(bounds, uniq_idx, side="right")          # rank holding each unique fragment
    first_in_blk = np.concatenate(([0], np.nonzero(np.diff(blk))[0] + 1)) if len(blk) else np.zeros(0, dtype=np.int64)
    blk_owner = owner[first_in_blk] if len(blk) else np.zeros(0, dtype=np.int64)
    if balance and nblk:
        blk_owner = (np.arange(nblk, dtype=np.int64) * len(counts)) // nblk
    starts = np.concatenate((first_in_blk, [len(uniq_idx)])).astype(np.int64)
    # where every fragment of a block is read from: the rank of its first occurrence, or -- local_copies -- the block's owner
    # itself when it holds a duplicate
    src_rank = owner.copy()
    src_idx = uniq_idx.copy()
    if (balance if local_copies is None else local_copies) and len(blk):
        lo = bounds - np.asarray(counts, dtype=np.int64)
        for r in range(len(counts)):
            ks = np.nonzero((blk_owner[blk] == r) & (owner != r))[0]
            if not len(ks) or bounds[r] == lo[r]:
                continue
            uq, at = np.unique(first[lo[r]:bounds[r]], return_index=True)      # first occurrences rank r holds a copy of, and where
            pos = np.minimum(np.searchsorted(uq, uniq_idx[ks]), len(uq) - 1)
            has = uq[pos] == uniq_idx[ks]
            src_rank[ks[has]] = r
            src_idx[ks[has]] = lo[r] + at[pos[has]]
    mine = np.nonzero(blk_owner == rank)[0]
    blocks = {}
    recv = {}
    for b in mine:
        sl = slice(starts[b], starts[b + 1])
        blocks[int(b)] = (uniq_idx[sl], lens[uniq_idx[sl]], src_rank[sl], src_idx[sl])
        for src in np.unique(src_rank[sl]):
            if src != rank:
                recv.setdefault(int(src), []).append(uniq_idx[sl][src_rank[sl] == src])
    send = {}
    if len(blk):
        theirs = np.nonzero((src_rank == rank) & (blk_owner[blk] != rank))[0]
        for dst in np.unique(blk_owner[blk[theirs]]):
            send[int(dst)] = uniq_idx[theirs[blk_owner[blk[theirs]] == dst]]
    recv = {k: np.concatenate(v) for k, v in recv.items()}
    return dict(uniq_idx=uniq_idx, nblocks=int(nblk), mine=mine, blocks=blocks, send=send, recv=recv,
                starts=starts, owner=owner, blk=blk, blk_owner=blk_owner, is_new=is_new)


def first_occurrence(digests20):
    """Exact first-occurrence index over an (n,20) uint8 digest table (CPU stand-in for zpq_dedup_dev in tests)."""
    d = np.ascontiguousarray(digests20).view([("k", "V20")]).ravel()
    _, idx, inv = np.unique(d, return_index=True, return_inverse=True)
    return idx[inv].astype(np.int64)
