#!/usr/bin/env python3
"""Model of the GPU suffix-array construction (zpaqfranz_amd/csrc/lz77_sa.hip): prefix doubling with discarding, in the
same steps as the kernels -- 8-byte keys first, then per round (dense group id, rank of the suffix h further on) keys
sorted for the still ambiguous suffixes only, new group heads, singleton compaction.  numpy stands in for the device
primitives (radix sort of pairs, scans).  tests/test_sa_cpu.py checks it against the oracle."""
import numpy as np


def suffix_array(data):
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    n = len(b)
    if n == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0
    pad = np.concatenate([b, np.zeros(8, np.uint8)]).astype(np.uint64)
    key = np.zeros(n, np.uint64)
    for k in range(8):
        key = (key << np.uint64(8)) | pad[k:k + n]
    val = np.arange(n, dtype=np.int64)
    pos = np.arange(n, dtype=np.int64)
    sa = np.zeros(n, np.int64)
    rank = np.zeros(n, np.int64)
    h = 8
    rounds = 0
    while True:
        order = np.argsort(key, kind="stable")          # device: radix sort of (key, val)
        key, val = key[order], val[order]
        m = len(key)
        head = np.ones(m, bool)
        head[1:] = key[1:] != key[:-1]
        gstart = np.maximum.accumulate(np.where(head, pos, 0))   # max-scan: slot of the group's first element
        sa[pos] = val
        rank[val] = gstart
        nxt = np.ones(m, bool)
        nxt[:-1] = head[1:]
        keep = ~(head & nxt)                             # singletons are final
        rounds += 1
        if not keep.any():
            break
        gid = np.cumsum(head & keep)[keep] - 1           # dense group number among the kept
        pos, val = pos[keep], val[keep]
        ih = val + h
        inr = ih < n
        k2 = np.where(inr, rank[np.minimum(ih, n - 1)] + h + 1, n - val)   # beyond the end: the shorter suffix first
        b2 = int(n + h + 1).bit_length()
        key = (gid.astype(np.uint64) << np.uint64(b2)) | k2.astype(np.uint64)
        h *= 2
    return sa.astype(np.uint32), rank.astype(np.uint32), rounds


if __name__ == "__main__":
    import sys
    d = open(sys.argv[1], "rb").read()
    sa, isa, r = suffix_array(d)
    print(len(sa), r)
