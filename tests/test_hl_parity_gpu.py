"""Whole-instance parity on the MI355X: the host conflict-tree drivers (libmrp_hl.so) calling the HIP low-level engine
must reproduce the oracle's `statistics:` block (cost, makespan, highLevelExpanded, lowLevelExpanded) and every path.

* reference known answers (test/test_cbs.py:24-34, test/test_ecbs.py:25-35) on the reference's own fixtures;
* shipped benchmark inputs vs oracle_expected.json (produced by the oracle, see tests/golden/make_fixtures.py);
* seeded synthetic instances (the bench workload) vs the oracle run live.
"""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(("|" + ",".join("%d:%d" % (x, y) for x, y in p)).encode())
    return h.hexdigest()[:16]


@pytest.fixture(scope="module")
def solver():
    from libmultirobotplanning_amd import hl
    s = hl.BatchSolver(device=0, n_threads=4, slots=512)
    yield s
    s.close()


def test_reference_known_answers(solver, ref_tests):
    from libmultirobotplanning_amd import hl
    names = ["mapf_simple1", "mapf_circle", "mapf_atGoal"]
    insts = [ref_tests["mapf"][n] for n in names]
    res, _ = solver.solve(insts, algo=hl.CBS)
    assert [r["cost"] for r in res] == [ref_tests["cbs_cost"][n] for n in names]      # test/test_cbs.py:24-34
    assert all(r["status"] == hl.SOLVED for r in res)
    res, _ = solver.solve(insts, algo=hl.ECBS, w=1.0)
    assert [r["cost"] for r in res] == [ref_tests["ecbs_w1_cost"][n] for n in names]  # test/test_ecbs.py:25-35


def test_ecbs_w13_benchmark_instances(solver, bench_instances, oracle_expected):
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "32by32" in n]
    names = [n for n in names if oracle_expected[n]["ecbs_w1.3"]["rc"] == 1]
    assert len(names) >= 140
    res, stats = solver.solve([bench_instances[n] for n in names], algo=hl.ECBS, w=1.3)
    for n, r in zip(names, res):
        e = oracle_expected[n]["ecbs_w1.3"]
        assert r["status"] == hl.SOLVED, n
        assert (r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (e["cost"], e["makespan"], e["hl"],
                                                                                 e["ll"]), n
        assert _digest(r["paths"]) == e["digest"], n
    assert stats["solved"] == len(names)
    assert stats["ll_expansions"] == sum(r["ll_expanded"] for r in res)


def test_rounds_mode_gives_the_same_results(solver, bench_instances, oracle_expected):
    """mode=1 (one launch per round) and the default session mode are two schedules of the same searches."""
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "32by32" in n and ("agents10_" in n or "agents50_" in n)][:60]
    res, _ = solver.solve([bench_instances[n] for n in names], algo=hl.ECBS, w=1.3, mode=1)
    for n, r in zip(names, res):
        e = oracle_expected[n]["ecbs_w1.3"]
        assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"], _digest(r["paths"])) == (
            hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"], e["digest"]), n


def test_cbs_and_ecbs_8x8(solver, bench_instances, oracle_expected):
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "8by8" in n]
    cbs_names = [n for n in names if oracle_expected[n]["cbs"]["rc"] == 1]
    res, _ = solver.solve([bench_instances[n] for n in cbs_names], algo=hl.CBS)
    for n, r in zip(cbs_names, res):
        e = oracle_expected[n]["cbs"]
        assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
            hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"]), n
        assert _digest(r["paths"]) == e["digest"], n
    # ECBS on every 8x8 fixture up to 20 agents, with the cap the vectors were produced with: instances the oracle could
    # not finish must come back as CAP, the others exactly
    res, _ = solver.solve([bench_instances[n] for n in names], algo=hl.ECBS, w=1.3, max_ll_expansions=300_000)
    for n, r in zip(names, res):
        e = oracle_expected[n]["ecbs_w1.3"]
        if e["rc"] == 1:
            assert (r["status"], r["cost"], r["hl_expanded"], r["ll_expanded"], _digest(r["paths"])) == (
                hl.SOLVED, e["cost"], e["hl"], e["ll"], e["digest"]), n
        else:
            assert r["status"] == hl.CAP, n


def test_cbs_8x8_up_to_20_agents(solver, bench_instances, oracle_expected):
    """BASELINE.json configs[2]: CBS on benchmark/8x8_obst12 with agents up to 20, under the cap of the golden vectors
    (300 000 low-level expansions per instance).  Solved on the CPU => identical on the GPU; capped there => CAP here."""
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "8by8" in n and
             int(n.split("agents")[1].split("_")[0]) >= 10]
    assert {int(n.split("agents")[1].split("_")[0]) for n in names} == {10, 12, 16, 20}
    res, _ = solver.solve([bench_instances[n] for n in names], algo=hl.CBS, max_ll_expansions=300_000)
    n_solved = 0
    for n, r in zip(names, res):
        e = oracle_expected[n]["cbs"]
        if e["rc"] == 1:
            n_solved += 1
            assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"], _digest(r["paths"])) == (
                hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"], e["digest"]), n
        else:
            assert r["status"] == hl.CAP, n
            assert r["ll_expanded"] > 300_000
    assert n_solved >= 5


def test_caps_are_reported(solver, bench_instances, oracle_expected):
    """Instances the oracle could not finish under its cap come back as CAP here too — never as a wrong answer."""
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "8by8" in n and oracle_expected[n]["cbs"]["rc"] == -1][:3]
    res, _ = solver.solve([bench_instances[n] for n in names], algo=hl.CBS, max_ll_expansions=20_000)
    assert all(r["status"] == hl.CAP for r in res)


def test_synthetic_bench_workload_matches_oracle(solver, oracle_mod):
    from libmultirobotplanning_amd import hl
    insts = [hl.generate_instance(1000 * 10 + k, 32, 32, 204, 10) for k in range(48)]
    insts += [hl.generate_instance(1000 * 40 + k, 32, 32, 204, 40) for k in range(6)]
    res, stats = solver.solve(insts, algo=hl.ECBS, w=1.3)
    for inst, r in zip(insts, res):
        o = oracle_mod.mapf_solve(oracle_mod.ECBS, inst, w=1.3, cap_total=5_000_000)
        assert o["rc"] == 1 and r["status"] == hl.SOLVED
        assert (r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
            o["cost"], o["makespan"], o["hl_expanded"], o["ll_expanded"])
        assert r["paths"] == o["paths"]
    # size-independent property of any solution: no vertex / swap conflict before the reference's scan horizon
    for r in res:
        paths = r["paths"]
        T = max(len(p) for p in paths) - 1
        at = lambda p, t: tuple(p[min(t, len(p) - 1)])
        for t in range(T):
            cells = [at(p, t) for p in paths]
            assert len(set(cells)) == len(cells)
            nxt = [at(p, t + 1) for p in paths]
            for i in range(len(paths)):
                for j in range(i + 1, len(paths)):
                    assert not (cells[i] == nxt[j] and nxt[i] == cells[j])


def test_prioritized_sipp_known_answers_and_64x64(solver, oracle_mod, ref_tests):
    """example/mapf_prioritized_sipp.cpp on the GPU: the reference's own fixtures (test_mapf_prioritized_sipp.py:24-52)
    and config 5's shape (64x64, 10 % obstacles) against the oracle — cost, planned flags and every schedule."""
    from libmultirobotplanning_amd import hl
    exp = ref_tests["prioritized_sipp"]
    names = list(exp["cost"].keys())
    res, _ = solver.prioritized_sipp([ref_tests["mapf"][n] for n in names])
    assert [r["cost"] for r in res] == [exp["cost"][n] for n in names]
    b = res[names.index("mapf_simple1b")]
    assert len(b["schedules"][0]) == exp["simple1b_lens"]["agent0"] and len(b["schedules"][1]) == 0
    insts = [hl.generate_instance(64000 + k, 64, 64, 410, 60) for k in range(6)]
    insts += [hl.generate_instance(32000 + k, 32, 32, 204, 100) for k in range(4)]
    res, stats = solver.prioritized_sipp(insts)
    for inst, r in zip(insts, res):
        o = oracle_mod.prioritized_sipp(inst)
        assert (r["cost"], r["planned"], r["expanded"]) == (o["cost"], o["planned"], o["expanded"])
        assert r["schedules"] == o["schedules"]
    assert stats["rounds"] == 100


@pytest.mark.parametrize("batch_mode", [False, True])
def test_prioritized_sipp_capacity_status_is_per_instance(solver, oracle_mod, ref_tests, batch_mode):
    """A search that ends with a capacity status (here: the expansion cap of the test knob) stops ITS instance — status
    says which, the agents planned before it keep the reference's schedules — and leaves every other instance of the
    batch alone (round 2 failed the whole call); both schedules of the driver."""
    import os
    from libmultirobotplanning_amd import hl, ll
    names = list(ref_tests["prioritized_sipp"]["cost"].keys())
    insts = [ref_tests["mapf"][n] for n in names] + [hl.generate_instance(32000 + k, 32, 32, 204, 30) for k in range(6)]
    knobs = {"MRP_HL_SIPP_MAX_EXPANSIONS": "60"}
    if batch_mode:
        knobs["MRP_HL_SIPP_BATCH"] = "1"
    os.environ.update(knobs)
    try:
        res, _ = solver.prioritized_sipp(insts)
    finally:
        for k in knobs:
            del os.environ[k]
    stopped = 0
    for inst, r in zip(insts, res):
        o = oracle_mod.prioritized_sipp(inst)
        if r["status"] == 0:
            assert (r["cost"], r["planned"], r["schedules"]) == (o["cost"], o["planned"], o["schedules"])
            continue
        stopped += 1
        assert r["status"] == ll.CAP_EXPANSIONS
        # the prefix planned before the capped search is the reference's; nothing is planned after it
        k = next((a for a in range(len(r["planned"]))
                  if r["planned"][a] != o["planned"][a] or r["schedules"][a] != o["schedules"][a]), len(r["planned"]))
        assert r["planned"][k:] == [0] * (len(r["planned"]) - k)
        assert r["schedules"][:k] == o["schedules"][:k] and r["n_planned"] == sum(o["planned"][:k])
    assert 0 < stopped < len(insts)


def test_root_chains_change_nothing_but_the_number_of_jobs(solver, oracle_mod, bench_instances):
    """MRP_LL_JOB_ROOT_CHAIN (the root step of an ECBS tree as one job, ecbs.hpp:118-136): same results as one job per root
    search — every counter, every path — on shipped inputs (whose long root searches break chains and resume them) and
    synthetic ones, against the oracle and against the driver with chains switched off; far fewer tickets."""
    import os
    from libmultirobotplanning_amd import hl
    insts = [bench_instances["map_32by32_obst204_agents%d_ex%d" % (a, k)] for a, k in ((10, 13), (10, 0), (20, 3), (30, 2), (30, 7))]
    insts += [hl.generate_instance(1000 * 10 + 91000 + k, 32, 32, 204, 10) for k in range(600)]
    insts += [hl.generate_instance(1000 * 24 + 300 + k, 32, 32, 204, 24) for k in range(40)]
    res_on, st_on = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    os.environ["MRP_HL_ROOT_CHAIN"] = "0"
    try:
        res_off, st_off = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    finally:
        del os.environ["MRP_HL_ROOT_CHAIN"]
    assert st_on["rounds"] * 3 < st_off["rounds"]  # the chains were really used
    keys = ("status", "cost", "makespan", "hl_expanded", "ll_expanded", "ll_searches", "paths")
    for i, (a, b) in enumerate(zip(res_on, res_off)):
        assert [a.get(k) for k in keys] == [b.get(k) for k in keys], i
    # the root step in jobs of bounded length (mrp_ll_job.chain_count; the drivers' rule from 64 agents on, forced here)
    os.environ["MRP_HL_CHAIN_CHUNK"], os.environ["MRP_HL_CHAIN_CHUNK_FROM"] = "4", "2"
    try:
        res_ch, st_ch = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    finally:
        del os.environ["MRP_HL_CHAIN_CHUNK"], os.environ["MRP_HL_CHAIN_CHUNK_FROM"]
    assert st_on["rounds"] < st_ch["rounds"] < st_off["rounds"]
    for i, (a, b) in enumerate(zip(res_ch, res_off)):
        assert [a.get(k) for k in keys] == [b.get(k) for k in keys], i
    for inst, r in list(zip(insts, res_on))[:40]:
        o = oracle_mod.mapf_solve(oracle_mod.ECBS, inst, w=1.3, cap_total=200000)
        assert o["rc"] == 1 and (r["status"], r["cost"], r["hl_expanded"], r["ll_expanded"]) == (
            hl.SOLVED, o["cost"], o["hl_expanded"], o["ll_expanded"])


def test_conflict_free_roots_are_written_without_a_conflict_tree(solver, oracle_mod):
    """The workgroup that runs a root chain also scans the root solution for conflicts (getFirstConflict ecbs.cpp:401-452 +
    focalHeuristic :315-350, mrp_ll.h MRP_LL_JOB_ROOT_CHAIN); a conflict-free root is written out by the driver without
    building a conflict tree.  Same results — every counter, every path, the schedule digest — as with the fast path
    switched off, and as the oracle; about seven ten-agent instances in ten take it."""
    import os
    from libmultirobotplanning_amd import hl
    insts = [hl.generate_instance(1000 * 10 + 55000 + k, 32, 32, 204, 10) for k in range(1500)]
    insts += [hl.generate_instance(1000 * 4 + 500 + k, 8, 8, 12, 4) for k in range(200)]   # tiny maps: many conflicts
    insts += [hl.generate_instance(1000 * 28 + 900 + k, 32, 32, 204, 28) for k in range(40)]
    res_on, st_on = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    os.environ["MRP_HL_ROOT_FAST"] = "0"
    try:
        res_off, st_off = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    finally:
        del os.environ["MRP_HL_ROOT_FAST"]
    assert st_off["root_solved"] == 0 and st_on["root_solved"] > 900
    # (a root chain that stopped in front of a search too large for the LDS tier is resumed behind it: no scan, usual path)
    n_hl1 = sum(1 for r in res_on if r["status"] == hl.SOLVED and r["hl_expanded"] == 1 and len(r["paths"]) >= 2)
    assert n_hl1 * 0.97 <= st_on["root_solved"] <= n_hl1
    keys = ("status", "cost", "makespan", "hl_expanded", "ll_expanded", "ll_searches", "paths", "schedule_digest")
    for i, (a, b) in enumerate(zip(res_on, res_off)):
        assert [a.get(k) for k in keys] == [b.get(k) for k in keys], i
    for inst, r in list(zip(insts, res_on))[:300] + list(zip(insts, res_on))[1500:1600]:
        o = oracle_mod.mapf_solve(oracle_mod.ECBS, inst, w=1.3, cap_total=200000, path_cap=1024)
        assert o["rc"] == 1 and (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"], r["paths"]) == (
            hl.SOLVED, o["cost"], o["makespan"], o["hl_expanded"], o["ll_expanded"], o["paths"])


def test_ecbs_without_the_compact_tier(oracle_mod, bench_instances):
    """mrp_ll_options.lds_nodes < 0 switches the compact tier off; results never depend on it (mrp_ll.h).  The engine then
    cannot run root chains either: it rejects the first one and the driver goes on with one job per root search."""
    from libmultirobotplanning_amd import hl
    insts = [bench_instances["map_32by32_obst204_agents10_ex%d" % k] for k in range(6)]
    insts += [hl.generate_instance(1000 * 10 + 77000 + k, 32, 32, 204, 10) for k in range(60)]
    s = hl.BatchSolver(device=0, n_threads=2, slots=128, lds_nodes=-1)
    try:
        res, st = s.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=200000)
    finally:
        s.close()
    for inst, r in zip(insts, res):
        o = oracle_mod.mapf_solve(oracle_mod.ECBS, inst, w=1.3, cap_total=200000)
        assert o["rc"] == 1 and (r["status"], r["cost"], r["hl_expanded"], r["ll_expanded"]) == (
            hl.SOLVED, o["cost"], o["hl_expanded"], o["ll_expanded"])


def test_opt_in_memory_placements_give_the_same_results(oracle_mod, ref_tests):
    """MRP_LL_RING_IN_DEVICE=1 (the host-to-device half of the job ring in uncached device memory, written through the BAR;
    off by default): same answers, ECBS and prioritized SIPP."""
    import os
    from libmultirobotplanning_amd import hl
    insts = [hl.generate_instance(1000 * 10 + 55000 + k, 32, 32, 204, 10) for k in range(512)]
    sipp = [hl.generate_instance(64000 + 100 + k, 64, 64, 410, 40) for k in range(24)]
    base = hl.BatchSolver(device=0, n_threads=4)
    try:
        want, _ = base.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
        want_sipp, _ = base.prioritized_sipp(sipp)
    finally:
        base.close()
    os.environ["MRP_LL_RING_IN_DEVICE"] = "1"
    try:
        s = hl.BatchSolver(device=0, n_threads=4)
        try:
            got, _ = s.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
            got_sipp, _ = s.prioritized_sipp(sipp)
        finally:
            s.close()
    finally:
        del os.environ["MRP_LL_RING_IN_DEVICE"]
    assert got == want
    assert got_sipp == want_sipp
    o = oracle_mod.prioritized_sipp(sipp[0])
    assert (got_sipp[0]["cost"], got_sipp[0]["planned"], got_sipp[0]["schedules"]) == (o["cost"], o["planned"], o["schedules"])


def test_bench_scale_properties_and_determinism(solver):
    """At the bench workload's size the oracle is too slow to check everything, so size-independent properties are
    checked on 4096 synthetic agents10 instances (and the oracle on a sample):
      * two runs (different thread counts => different schedules of the same searches) give identical results;
      * every returned schedule is valid: starts/goals right, unit moves on free cells, and no vertex / swap conflict
        before the reference's scan horizon (getFirstConflict, example/ecbs.cpp:401-452);
      * cost == sum of path lengths >= sum of Manhattan distances; makespan == longest path."""
    import numpy as np
    from libmultirobotplanning_amd import hl
    insts = [hl.generate_instance(1000 * 10 + 77000 + k, 32, 32, 204, 10) for k in range(4096)]
    res_a, st_a = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
    res_b, st_b = solver.solve(insts, algo=hl.ECBS, w=1.3, max_ll_expansions=50000, n_threads=3)
    assert st_a["ll_expansions"] == st_b["ll_expansions"]
    # ... and so do the fixed interleaved split of the instances over the workers (instead of the shared pool) and a
    # tight admission limit (instances wait in the pool; the pool is refilled as they finish)
    import os
    for knob, val in (("MRP_HL_STATIC_SPLIT", "1"), ("MRP_HL_ACTIVE_LIMIT", "37")):
        os.environ[knob] = val
        try:
            res_c, st_c = solver.solve(insts[:1024], algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
        finally:
            del os.environ[knob]
        assert res_c == res_a[:1024], knob
    n_solved = 0
    for inst, a, b in zip(insts, res_a, res_b):
        assert a == b
        if a["status"] != hl.SOLVED:
            assert a["status"] == hl.CAP
            continue
        n_solved += 1
        obst = {tuple(o) for o in inst["obstacles"]}
        paths = a["paths"]
        assert a["cost"] == sum(len(p) - 1 for p in paths)
        assert a["makespan"] == max(len(p) - 1 for p in paths)
        lower = 0
        for p, s, g in zip(paths, inst["starts"], inst["goals"]):
            assert p[0] == s and p[-1] == g
            lower += abs(s[0] - g[0]) + abs(s[1] - g[1])
            for (x0, y0), (x1, y1) in zip(p, p[1:]):
                assert abs(x0 - x1) + abs(y0 - y1) <= 1
                assert 0 <= x1 < 32 and 0 <= y1 < 32 and (x1, y1) not in obst
        assert a["cost"] >= lower
        T = max(len(p) for p in paths) - 1
        arr = np.array([[p[min(t, len(p) - 1)] for t in range(T + 1)] for p in paths])  # [agent, t, 2]
        cell = arr[:, :, 1] * 32 + arr[:, :, 0]
        for t in range(T):
            col = cell[:, t]
            assert len(np.unique(col)) == len(col)
            nxt = cell[:, t + 1]
            swap = (col[:, None] == nxt[None, :]) & (nxt[:, None] == col[None, :])
            np.fill_diagonal(swap, False)
            assert not swap.any()
    assert n_solved >= 4090


@pytest.mark.gpu
def test_cbs_stream_on_the_gpu(solver, bench_instances, oracle_expected):
    """mrp_hl_solver_solve_stream with CBS sessions (mrp_ll_cbs_persistent_kernel): the shipped 8x8 inputs the oracle solves,
    as three batches (the middle one empty) of one stream, against the golden vectors — schedules included."""
    from libmultirobotplanning_amd import hl
    names = [n for n in sorted(bench_instances) if "8by8" in n and oracle_expected[n]["cbs"]["rc"] == 1]
    groups = [names[:7], [], names[7:]]
    preps = [solver.prepare([bench_instances[n] for n in g], want_paths=True, path_cap=256) for g in groups]
    try:
        st = solver.solve_stream(preps, algo=hl.CBS)
        for g, prep in zip(groups, preps):
            for n, r in zip(g, solver.results_of(prep)):
                e = oracle_expected[n]["cbs"]
                assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
                    hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"]), n
                assert _digest(r["paths"]) == e["digest"], n
        assert st["solved"] == len(names)
    finally:
        for prep in preps:
            solver.release(prep)


@pytest.mark.gpu
def test_stream_of_batches_equals_separate_solves_on_the_gpu(solver):
    """mrp_hl_solver_solve_stream on the GPU: four synthetic agents10 batches as one stream (no barrier between batches)
    against the same batches solved one call each — status of every instance; cost, both expansion counters, schedule digest
    and path lengths of every solved one."""
    import numpy as np
    from libmultirobotplanning_amd import hl
    batches = [hl.generate_instances(77000 + 5000 * b, n, 32, 32, 204, 10) for b, n in enumerate((3000, 1, 2048, 777))]
    a = [solver.prepare(b, want_paths=True, path_cap=128) for b in batches]
    c = [solver.prepare(b, want_paths=True, path_cap=128) for b in batches]
    try:
        st = solver.solve_stream(a, algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
        n_solved = 0
        for pa, pc in zip(a, c):
            _, s1 = solver.solve_prepared(pc, algo=hl.ECBS, w=1.3, max_ll_expansions=50000, raw=True)
            n_solved += s1["solved"]
            ra, rc = solver.result_arrays(pa), solver.result_arrays(pc)
            solved = rc["status"] == hl.SOLVED
            assert np.array_equal(ra["status"], rc["status"])
            for f in ("cost", "makespan", "hl_expanded", "ll_expanded", "schedule_digest"):
                assert np.array_equal(ra[f][solved], rc[f][solved]), f
            assert np.array_equal(ra["path_len"][solved], rc["path_len"][solved])
        # (the counters of an instance that ends at the harness cap depend on the look-ahead, i.e. on the schedule: DESIGN §4)
        assert st["solved"] == n_solved and st["batches"] == 4
    finally:
        for p in a + c:
            solver.release(p)


def test_caller_stepped_conflict_tree_on_the_gpu(bench_instances, oracle_expected):
    """mrp_hl_ct_* + ct_sharded.solve_sharded with the product executor (this GPU through the C-ABI), world size 1: the
    same results as the batch drivers and the oracle for look-ahead widths 1 and 4 (the multi-rank exchange itself is
    covered by tests/test_sharding_gloo.py)."""
    from libmultirobotplanning_amd import ct_sharded, hl
    for n, algo, key in (("map_32by32_obst204_agents50_ex1", hl.ECBS, "ecbs_w1.3"), ("map_8by8_obst12_agents8_ex3", hl.CBS, "cbs")):
        inst = bench_instances[n]
        run = ct_sharded.gpu_executor(inst, device=0)
        try:
            for k in (1, 4):
                r = ct_sharded.solve_sharded(inst, run, None, algo=algo, w=1.3, spec_width=k)
                e = oracle_expected[n][key]
                assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"], _digest(r["paths"])) == (
                    hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"], e["digest"]), (n, k)
        finally:
            run.close()


def test_all_1000_shipped_32x32_inputs(solver, shipped_corpus):
    """The corpus BASELINE.json's north_star names, whole: every shipped benchmark/32x32_obst204 input (agents10..100 x
    ex0..99) as one batch against tests/golden/shipped_32x32_expected.json (our oracle, cap 3 000 000: agents100_ex36 runs
    into it on both sides)."""
    from libmultirobotplanning_amd import hl
    corpus, exp = shipped_corpus
    assert len(corpus) == 1000
    res, stats = solver.solve([i for _, i in corpus], algo=hl.ECBS, w=1.3, max_ll_expansions=3_000_000, path_cap=1024)
    capped = 0
    for (n, _), r in zip(corpus, res):
        e = exp[n]
        if e["rc"] == 1:
            assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
                hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"]), n
            assert _digest(r["paths"]) == e["digest"], n
        else:
            capped += 1
            assert r["status"] == hl.CAP, n
    assert capped == 1 and exp["map_32by32_obst204_agents100_ex36"]["rc"] != 1


@pytest.mark.timeout(900)
def test_the_heavy_tail_input_runs_to_completion(solver, shipped_corpus):
    """benchmark/32x32_obst204/map_32by32_obst204_agents100_ex36 with NO cap, as the reference runs it (a_star_epsilon.hpp:116
    and ecbs.hpp:151 have none): one conflict tree of 70 612 nodes, 56.8 million low-level expansions — SURVEY.md §6's
    figures from the reference's own headers, reproduced by the uncapped oracle vector
    tests/golden/shipped_heavy_tail_expected.json — about a minute on the GPU (one dependent chain of rounds)."""
    import json
    import os
    from libmultirobotplanning_amd import hl
    with open(os.path.join(os.path.dirname(__file__), "golden", "shipped_heavy_tail_expected.json")) as f:
        e = json.load(f)["map_32by32_obst204_agents100_ex36"]
    assert (e["cost"], e["hl"], e["ll"]) == (2574, 70612, 56795846)
    corpus, _ = shipped_corpus
    inst = dict(corpus)["map_32by32_obst204_agents100_ex36"]
    res, _ = solver.solve([inst], algo=hl.ECBS, w=1.3, max_ll_expansions=-1, path_cap=1024)
    r = res[0]
    assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
        hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"])
    assert _digest(r["paths"]) == e["digest"]


def test_cbs_8x8_at_the_surveys_cap(solver, bench_instances):
    """CBS on shipped 8x8_obst12 inputs with a cap of 1 000 000 low-level expansions per instance —
    agents10 / 12 inputs that need hundreds of thousands of expansions get real parity, agents16 / 20 cap on both sides."""
    import json
    import os
    from libmultirobotplanning_amd import hl
    with open(os.path.join(os.path.dirname(__file__), "golden", "cbs_8x8_cap1e6.json")) as f:
        exp = json.load(f)
    names = sorted(exp)
    res, _ = solver.solve([bench_instances[n] for n in names], algo=hl.CBS, max_ll_expansions=1_000_000)
    solved = 0
    for n, r in zip(names, res):
        e = exp[n]
        if e["rc"] == 1:
            solved += 1
            assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"], _digest(r["paths"])) == (
                hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"], e["digest"]), n
        else:
            assert r["status"] == hl.CAP, n
    assert solved >= 6
