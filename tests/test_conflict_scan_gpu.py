"""SURVEY.md §8 f1 on the MI355X: the conflict-scan kernel (csrc/conflict_kernel.hip, mrp_ll_conflict_scan) against the
oracle's restatement of Environment::getFirstConflict (example/ecbs.cpp:401-452) and Environment::focalHeuristic
(:315-350): first conflict in the reference's scan order (t, vertex before edge, i, j; the last time step is never
checked) and the number of all conflicts."""
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from libmultirobotplanning_amd import ll
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=16)
    yield eng
    eng.close()


def _random_walk(rng, dim, length):
    x, y = rng.randrange(dim), rng.randrange(dim)
    p = [[x, y]]
    for _ in range(length - 1):
        dx, dy = rng.choice(((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)))
        x, y = min(max(x + dx, 0), dim - 1), min(max(y + dy, 0), dim - 1)
        p.append([x, y])
    return p


def test_random_path_sets(engine, oracle_mod):
    """12 000 collision-rich random solutions (2..12 agents on 3x3..6x6 grids, ragged path lengths including single-state
    paths) plus 60 large ones (65..150 agents: several 64-lane chunks, 256x256 coordinates)."""
    rng = random.Random(2024)
    sets = []
    for _ in range(12000):
        dim = rng.randrange(3, 7)
        sets.append([_random_walk(rng, dim, rng.randrange(1, 13)) for _ in range(rng.randrange(2, 13))])
    for _ in range(60):
        dim = rng.choice((12, 40, 256))
        sets.append([_random_walk(rng, dim, rng.randrange(1, 40)) for _ in range(rng.randrange(65, 151))])
    sets.append([[[3, 3]]])                      # one agent: nothing to compare
    sets.append([[[0, 0]], [[0, 0]]])            # same cell, but the only time step is the last one: never checked
    sets.append([[[0, 0], [0, 0]], [[0, 0]]])    # ... and checked as soon as any path is longer
    got = engine.conflict_scan(sets)
    n_found = n_edge = 0
    for sol, g in zip(sets, got):
        assert g == oracle_mod.conflict_scan(sol), sol
        n_found += g["found"]
        n_edge += g["type"]
    assert n_found > 5000 and n_edge > 300
    assert engine.conflict_scan([]) == []


def test_every_ct_node_of_the_agents100_fixtures(engine, oracle_mod, bench_instances):
    """The solution vectors of the conflict-tree nodes ECBS (w = 1.3) expands on the shipped agents100 inputs, harvested from
    the oracle's low-level calls (a child's focal context is its parent node's solution without the re-planned agent: 99
    paths of the real node), plus the agents50 ones."""
    names = [n for n in sorted(bench_instances) if "agents100_" in n][:6] + \
            [n for n in sorted(bench_instances) if "agents50_" in n][:6]
    sets = []
    for n in names:
        _, calls = oracle_mod.mapf_record(oracle_mod.ECBS, bench_instances[n], w=1.3, cap_total=3_000_000)
        n_agents = len(bench_instances[n]["starts"])
        for c in calls[n_agents:]:                 # the root's chain sees partial solutions: skip those
            others = [p for p in c["ctx_paths"] if len(p) > 0]   # the searching agent's own path is recorded empty
            assert len(others) >= n_agents - 1
            sets.append(others)
    assert len(sets) > 800
    got = engine.conflict_scan(sets)
    for sol, g in zip(sets, got):
        assert g == oracle_mod.conflict_scan(sol)
    assert sum(g["found"] for g in got) > len(got) // 2


def test_rejects_what_the_reference_asserts_on(engine):
    with pytest.raises(RuntimeError):
        engine.conflict_scan([[[[0, 0]], []]])     # getState asserts a non-empty path (ecbs.cpp:491)
    with pytest.raises(RuntimeError):
        engine.conflict_scan([[[[0, 300]], [[1, 1]]]])
