"""The multi-rank bench keeps several steps in flight on several threads; every rank must nevertheless issue its
collectives in one order (bench.CollectiveOrder).  Pure host logic: checked with threads and random delays."""
import os
import random
import sys
import threading
import time

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("steps,depth", [(1, 3), (2, 3), (7, 3), (15, 3), (5, 2), (4, 4), (3, 1)])
def test_every_section_gets_its_turn_in_the_fixed_order(steps, depth):
    from bench import CollectiveOrder
    o = CollectiveOrder(steps, depth)
    assert sorted(o.seq) == [(s, k) for s in range(steps) for k in range(3)]           # each section exactly once
    assert all(o.seq.index((s, 0)) < o.seq.index((s, 1)) < o.seq.index((s, 2)) for s in range(steps))
    assert o.seq == CollectiveOrder(steps, depth).seq                                   # a function of (steps, depth) only
    nxt, lock, log = [0], threading.Lock(), []

    def worker():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= steps:
                return
            for sec in range(3):
                time.sleep(random.random() * 0.002)
                o.enter(i, sec); log.append((i, sec)); o.leave()
    th = [threading.Thread(target=worker, daemon=True) for _ in range(depth)]
    for t in th: t.start()
    for t in th: t.join(20)
    assert not any(t.is_alive() for t in th), "deadlock"
    assert log == o.seq


def test_four_sections_per_step_for_the_product_call():
    """zpqj_add_sharded_dev has two late collectives per add (block sizes as a host string, the blocks through the device form)"""
    from bench import CollectiveOrder
    o = CollectiveOrder(5, 3, 4)
    assert sorted(o.seq) == [(s, k) for s in range(5) for k in range(4)]
    assert all(o.seq.index((s, 0)) < o.seq.index((s, 1)) < o.seq.index((s, 2)) < o.seq.index((s, 3)) for s in range(5))
    assert o.seq.index((1, 0)) < o.seq.index((0, 2))          # the next step's tables go out before this one's blocks
