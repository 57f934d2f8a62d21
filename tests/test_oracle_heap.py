"""Exact-heap restatement (oracle/heap_restated.hpp): structural invariants + the two ordered-walk forms agree."""
import random


def _check_heap(layout, keys):
    for i in range(1, len(layout)):
        assert not keys[layout[(i - 1) // 2]] < keys[layout[i]]


def test_heap_rules_small(oracle_mod):
    # push 5,5,5 then pop: boost siftdown moves the last element below equal children (first child wins ties)
    layout, w1, w2 = oracle_mod.heap_replay([[0, 5, 0], [0, 5, 0], [0, 5, 0]])
    assert layout == [0, 1, 2] and w1 == w2
    layout, _, _ = oracle_mod.heap_replay([[0, 5, 0], [0, 5, 0], [0, 5, 0], [1, 0, 0]])
    # pop: swap(front, back) -> [2,1]; siftdown(0): child 1 not less than 2 -> swap -> [1,2]
    assert layout == [1, 2]
    # erase of a middle handle bubbles it to the root unconditionally
    layout, _, _ = oracle_mod.heap_replay([[0, 9, 0], [0, 7, 0], [0, 8, 0], [0, 1, 0], [2, 3, 0]])
    assert sorted(layout) == [0, 1, 2]


def test_heap_random_ops_and_ordered_walks(oracle_mod):
    rng = random.Random(1234)
    for trial in range(200):
        ops, live, keys, nxt = [], [], {}, 0
        for _ in range(rng.randint(1, 120)):
            r = rng.random()
            if r < 0.55 or not live:
                k = rng.randint(0, 12)  # many ties on purpose
                ops.append([0, k, 0])
                # handle reuse mirrors the free list: last freed first
                live.append(None)
            elif r < 0.75:
                ops.append([1, 0, 0]); live.pop()
            else:
                ops.append([1, 0, 0]); live.pop()
        layout, w1, w2 = oracle_mod.heap_replay(ops)
        assert w1 == w2, trial
        assert sorted(w1) == sorted(layout)


def test_boost_crosscheck_hook_reports_absent_here(oracle_mod):
    """The oracle's heaps come from ORACLE_HEAP; `make liboracle_boost.so` rebuilds it on the real boost::heap::d_ary_heap
    where that header exists.  This image has no Boost: the hook must say so (and never pretend otherwise)."""
    import numpy as np
    assert oracle_mod.lib().oracle_heap_kind() == 0
    if oracle_mod.boost_lib() is None:
        per = np.zeros((1, 6), dtype=np.int64)
        assert oracle_mod.boost_crosscheck(oracle_mod.ECBS, 4, 4, np.zeros((1, 0, 2)), np.zeros((1, 1, 2)),
                                           np.zeros((1, 1, 2)), per) == "absent"
