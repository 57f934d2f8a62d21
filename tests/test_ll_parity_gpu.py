"""GPU parity of the HIP low-level searches (through the C-ABI, include/mrp_ll.h) against the oracle.

Every case runs the same low-level call — same map, agent, start/goal, constraint sets and (ECBS) CT-node paths — on
the MI355X and in the oracle's restatement of a_star.hpp / a_star_epsilon.hpp, and demands bit-exact equality of
success, cost, fmin, the onExpandNode count and the full path.  Calls are harvested from real CBS / ECBS runs of the
oracle on shipped benchmark inputs (tests/golden/bench_instances.json), so constraint sets and focal contexts are the
ones the conflict tree really produces.
"""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from libmultirobotplanning_amd import ll
    eng = ll.LowLevelEngine(device=0)
    yield eng
    eng.close()


def _harvest(oracle_mod, bench_instances, names, algo, w, cap_total):
    cases = []
    for name in names:
        inst = bench_instances[name]
        summary, calls = oracle_mod.mapf_record(algo, inst, w=w, cap_total=cap_total)
        for c in calls:
            cases.append((name, inst, c))
    return cases


def _run_and_compare(engine, cases, algo_ll, w, map_ids=None):
    from libmultirobotplanning_amd import ll
    map_ids = {} if map_ids is None else map_ids
    jobs = []
    for name, inst, c in cases:
        if name not in map_ids:
            map_ids[name] = engine.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        a = c["agent"]
        jobs.append(ll.LLJob(map_id=map_ids[name], algo=algo_ll, start=inst["starts"][a], goal=inst["goals"][a],
                             agent_idx=a, w=w, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]))
    res = engine.search_batch(jobs)
    assert len(res) == len(cases)
    for (name, inst, c), r in zip(cases, res):
        ctx = (name, c["agent"], len(c["vertex_constraints"]), len(c["edge_constraints"]))
        assert r.success == c["success"], ctx
        assert r.expanded == c["expanded"], ctx
        if c["success"]:
            assert r.status == ll.OK
            assert (r.cost, r.fmin) == (c["cost"], c["fmin"]), ctx
            assert [s[1:] for s in r.states] == c["states"], ctx
            assert [s[0] for s in r.states] == list(range(len(c["states"]))), ctx
        else:
            assert r.status == ll.NO_SOLUTION, ctx
    return res


def test_ecbs_low_level_calls_32x32(engine, oracle_mod, bench_instances):
    names = ["map_32by32_obst204_agents10_ex%d" % k for k in range(12)]
    names += ["map_32by32_obst204_agents20_ex0", "map_32by32_obst204_agents30_ex1", "map_32by32_obst204_agents50_ex0"]
    cases = _harvest(oracle_mod, bench_instances, names, oracle_mod.ECBS, 1.3, 3_000_000)
    assert len(cases) > 200
    from libmultirobotplanning_amd import ll
    _run_and_compare(engine, cases, ll.ASTAR_EPS, 1.3)


def test_ecbs_low_level_calls_100_agents(engine, oracle_mod, bench_instances):
    cases = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents100_ex0"], oracle_mod.ECBS, 1.3, 3_000_000)
    from libmultirobotplanning_amd import ll
    res = _run_and_compare(engine, cases, ll.ASTAR_EPS, 1.3)
    assert any(r.tier == 0 for r in res)  # (searches of thousands of expansions stay in the compact tier too)


def test_ecbs_w1_low_level_calls(engine, oracle_mod, bench_instances, ref_tests):
    from libmultirobotplanning_amd import ll
    cases = _harvest(oracle_mod, bench_instances, ["map_8by8_obst12_agents4_ex%d" % k for k in range(5)],
                     oracle_mod.ECBS, 1.0, 300_000)
    cases += _harvest(oracle_mod, ref_tests["mapf"], ["mapf_simple1", "mapf_circle", "mapf_atGoal", "mapf_swap4"],
                      oracle_mod.ECBS, 1.0, 300_000)
    _run_and_compare(engine, cases, ll.ASTAR_EPS, 1.0)


def test_cbs_low_level_calls_8x8(engine, oracle_mod, bench_instances, ref_tests):
    from libmultirobotplanning_amd import ll
    names = ["map_8by8_obst12_agents%d_ex%d" % (n, k) for n in (2, 4, 5, 6) for k in range(4)]
    cases = _harvest(oracle_mod, bench_instances, names, oracle_mod.CBS, 1.0, 300_000)
    cases += _harvest(oracle_mod, ref_tests["mapf"], ["mapf_simple1", "mapf_circle", "mapf_atGoal", "mapf_swap2"],
                      oracle_mod.CBS, 1.0, 300_000)
    assert len(cases) > 100
    _run_and_compare(engine, cases, ll.ASTAR, 1.0)


def test_tiers_agree(oracle_mod, bench_instances):
    """Same jobs through (a) a tiny LDS tier that forces migration, (b) the arena tier only — each with the default arena
    (131 072 nodes: 32-bit node ids in the heap entries) and with a 65 536-node arena (16-bit ids, the entries carry the
    node's cell; the conflict-tree drivers' default): identical results.  Agents50 jobs add heaps deeper than the part
    of them that lives in LDS."""
    from libmultirobotplanning_amd import ll
    cases = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents10_ex%d" % k for k in range(4)],
                     oracle_mod.ECBS, 1.3, 3_000_000)
    big = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents50_ex1"], oracle_mod.ECBS, 1.3, 3_000_000)
    for lds_nodes in (32, -1):
        for arena_nodes in (0, 65536):
            eng = ll.LowLevelEngine(device=0, lds_nodes=lds_nodes, arena_nodes=arena_nodes, n_tickets=1, slots=256)
            try:
                res = _run_and_compare(eng, cases, ll.ASTAR_EPS, 1.3)
                assert any(r.tier == 1 for r in res)
                res = _run_and_compare(eng, big, ll.ASTAR_EPS, 1.3)
                assert any(r.tier == 1 and r.expanded > 2000 for r in res)
            finally:
                eng.close()


def test_random_jobs_with_many_constraints(engine, oracle_mod):
    """Fuzz: random 16x16 maps, random vertex constraints, up to 150 edge constraints per job (more than one wave's
    worth: the kernel keeps 64 keys in a register and walks the rest), random focal contexts, both algorithms."""
    import random
    from libmultirobotplanning_amd import ll
    rng = random.Random(20260401)
    dim = 16
    moves = [(0, 0), (-1, 0), (1, 0), (0, 1), (0, -1)]
    total = 0
    for trial in range(6):
        obst = sorted({(rng.randrange(dim), rng.randrange(dim)) for _ in range(30)})
        obst = [list(o) for o in obst]
        free = [[x, y] for x in range(dim) for y in range(dim) if [x, y] not in obst]
        mid = engine.upload_map(dim, dim, obst)
        mp = dict(dimx=dim, dimy=dim, obstacles=obst)
        jobs, specs = [], []
        for case in range(40):
            st, go = rng.choice(free), rng.choice(free)
            vc = [[rng.randrange(0, 30)] + rng.choice(free) for _ in range(rng.randrange(0, 40))]
            ec = []
            for _ in range(rng.choice([0, 3, 20, 70, 150])):
                c = rng.choice(free)
                dx, dy = rng.choice(moves[1:])
                ec.append([rng.randrange(0, 30), c[0], c[1], c[0] + dx, c[1] + dy])
            ctx = []
            if case % 2:
                for a in range(rng.randrange(2, 9)):  # random walks as the other agents' paths; agent 0 is the searcher
                    p = [rng.choice(free)]
                    for _ in range(rng.randrange(0, 25)):
                        dx, dy = rng.choice(moves)
                        q = [p[-1][0] + dx, p[-1][1] + dy]
                        p.append(q if q in free else p[-1])
                    ctx.append(p if a else [])
            algo = ll.ASTAR_EPS if case % 2 else ll.ASTAR
            w = rng.choice([1.0, 1.3, 2.0]) if algo == ll.ASTAR_EPS else 1.0
            specs.append((algo, st, go, vc, ec, ctx, w))
            jobs.append(ll.LLJob(map_id=mid, algo=algo, start=st, goal=go, agent_idx=0, w=w, vertex_constraints=vc,
                                 edge_constraints=ec, ctx_paths=ctx, max_expansions=30000))
        res = engine.search_batch(jobs)
        for (algo, st, go, vc, ec, ctx, w), r in zip(specs, res):
            o = oracle_mod.ll_search(algo, mp, 0, st, go, vc, ec, ctx, w=w, cap_expansions=30000)
            if o["rc"] == -1:
                assert r.status == ll.CAP_EXPANSIONS
                continue
            assert r.success == o["success"], (trial, st, go)
            assert r.expanded == o["expanded"], (trial, st, go, len(ec))
            if o["success"]:
                assert (r.cost, r.fmin, r.states, r.actions) == (o["cost"], o["fmin"], o["states"], o["actions"])
            total += 1
    assert total > 150


def test_more_than_128_agents_in_the_focal_context(engine, oracle_mod):
    """150 agents: the path table rows are 160 entries wide, so the focal heuristics take the chunked branch of the
    kernel (columns >= 128) that the shipped benchmark sizes (<= 100 agents) never reach."""
    from libmultirobotplanning_amd import hl, ll
    inst = hl.generate_instance(424242, 32, 32, 204, 150)
    summary, calls = oracle_mod.mapf_record(oracle_mod.ECBS, inst, w=1.3, cap_total=200000)
    assert len(calls) > 300 and max(len(c["ctx_paths"]) for c in calls) == 150
    res = _run_and_compare(engine, [("synthetic150", inst, c) for c in calls], ll.ASTAR_EPS, 1.3)
    assert sum(r.expanded for r in res) == sum(c["expanded"] for c in calls)


def test_configure_tiers_changes_nothing_but_the_tier(oracle_mod, bench_instances):
    """mrp_ll_configure_tiers: any limits (open list nodes/2 entries, time steps) give the same bits — a search that
    outgrows them is run by the arena tier; occupancy follows the LDS bytes."""
    from libmultirobotplanning_amd import ll
    cases = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents10_ex%d" % k for k in range(10, 14)],
                     oracle_mod.ECBS, 1.3, 3_000_000)
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
    try:
        occ = []
        for geom in ((400, 48, 2048), (64, 16, 64), (1024, 96, 8192), (8, 8, 32)):
            occ.append(eng.configure_tiers(*geom))
            res = _run_and_compare(eng, cases, ll.ASTAR_EPS, 1.3)
            if geom[0] <= 64:
                assert any(r.tier == 1 for r in res)
        assert occ[0] >= 6 and occ[2] < occ[0]  # the tier's window is fixed (~18 KB); the path table comes on top
        eng.session_begin(64)
        try:
            with pytest.raises(RuntimeError):
                eng.configure_tiers(256, 32, 1024)  # MRP_LL_E_BUSY while a session is open
        finally:
            eng.session_end()
    finally:
        eng.close()


def test_session_mode_matches_oracle(oracle_mod, bench_instances):
    """Session mode (resident wavefronts fed through the pinned-host job ring): same jobs, same bits, any order."""
    from libmultirobotplanning_amd import ll
    cases = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents10_ex%d" % k for k in range(8)] +
                     ["map_32by32_obst204_agents50_ex1"], oracle_mod.ECBS, 1.3, 3_000_000)
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
    try:
        map_ids = {}
        for name, inst, _ in cases[:-40]:  # all but the last instance's map are known before the session starts
            if name not in map_ids:
                map_ids[name] = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        for rep in range(2):  # a second session on the same context must start clean
            eng.session_begin(128)
            try:
                for lo in range(0, len(cases), 500):  # batches smaller than the ring; one map is uploaded in-session
                    res = _run_and_compare(eng, cases[lo:lo + 500], ll.ASTAR_EPS, 1.3, map_ids)
                    assert any(r.tier == 0 for r in res)
            finally:
                eng.session_end()
        _run_and_compare(eng, cases[:300], ll.ASTAR_EPS, 1.3, map_ids)  # and batch mode still works afterwards
        # ring wrap-around: many more tickets than ring slots, consumed out of order through mrp_ll_poll_any
        import ctypes
        import numpy as np
        lib = ll.load_library()
        os.environ["MRP_LL_TICKET_RING"] = "256"  # smallest ticket rings (1792 / 256 entries): they wrap 2x / 7x below
        try:
            eng.session_begin(64)
        finally:
            del os.environ["MRP_LL_TICKET_RING"]
        try:
            small = cases[:40]
            total = 6000  # 2048 job slots, re-used from the free list in completion order
            inflight = {}
            done_buf = (ctypes.c_int32 * 256)()
            n_done = ctypes.c_int32(0)
            submitted = completed = 0
            keep = {}
            while completed < total:
                while submitted < total and len(inflight) < 300:
                    name, inst, c = small[submitted % len(small)]
                    a = c["agent"]
                    job = ll.LLJob(map_id=map_ids[name], algo=ll.ASTAR_EPS, start=inst["starts"][a],
                                   goal=inst["goals"][a], agent_idx=a, w=1.3,
                                   vertex_constraints=c["vertex_constraints"],
                                   edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"])
                    cj, cr, hold = eng._marshal([job], 512)
                    tk = ctypes.c_int32(-1)
                    rc = lib.mrp_ll_submit_lane(eng._h, submitted % 3 == 0, 1, cj, cr, ctypes.byref(tk))  # both lanes
                    if rc == -4:
                        break
                    assert rc == 0
                    inflight[tk.value] = (submitted % len(small), cr, hold, cj)
                    submitted += 1
                assert lib.mrp_ll_poll_any(eng._h, done_buf, 256, ctypes.byref(n_done)) == 0
                for q in range(n_done.value):
                    which, cr, hold, cj = inflight.pop(done_buf[q])
                    c = small[which][2]
                    assert cr[0].expanded == c["expanded"] and cr[0].cost == c["cost"], (which, completed)
                    completed += 1
        finally:
            eng.session_end()
    finally:
        eng.close()


def test_front_and_heavy_workgroups(oracle_mod, bench_instances):
    """mrp_ll_session_begin_tiers: front workgroups (LDS tier only) hand the searches that outgrow their tier to heavy
    workgroups (wide LDS tier: 3071 open entries, long horizons; arena tier behind it) through a device-side queue.
    Same bits as the oracle whichever workgroup ran a search; the wide tier really takes the big ones, and the long ones (a
    198-step path through a serpentine: its entries carry h instead of g); the MRP_LL_JOB_HEAVY hint changes nothing."""
    from libmultirobotplanning_amd import ll
    cases = _harvest(oracle_mod, bench_instances, ["map_32by32_obst204_agents10_ex%d" % k for k in range(4)] +
                     ["map_32by32_obst204_agents100_ex2", "map_32by32_obst204_agents100_ex5"], oracle_mod.ECBS, 1.3, 3_000_000)
    # a serpentine: walls on every second row with a gap at alternating ends
    obst = [[x, y] for y in range(1, 12, 2) for x in range(32) if x != (31 if (y // 2) % 2 == 0 else 0)]
    snake = dict(dimx=32, dimy=32, obstacles=obst, starts=[[0, 0]], goals=[[31 if 6 % 2 else 0, 12]])
    o = oracle_mod.ll_search(oracle_mod.ASTAR_EPS, snake, 0, snake["starts"][0], snake["goals"][0], [], [], [], w=1.3)
    assert o["success"] and o["cost"] > 130
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
    try:
        map_ids = {}
        for name, inst, _ in cases:
            if name not in map_ids:
                map_ids[name] = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        snake_id = eng.upload_map(32, 32, obst)
        eng.session_begin_tiers(96, 12)
        try:
            tiers = []
            for lo in range(0, len(cases), 400):
                res = _run_and_compare(eng, cases[lo:lo + 400], ll.ASTAR_EPS, 1.3, map_ids)
                tiers += [r.tier for r in res]
            assert tiers.count(0) > 100 and tiers.count(2) >= 5, (tiers.count(0), tiers.count(1), tiers.count(2))
            r = eng.search_batch([ll.LLJob(map_id=snake_id, algo=ll.ASTAR_EPS, start=snake["starts"][0], goal=snake["goals"][0],
                                           w=1.3)])[0]
            assert (r.status, r.cost, r.fmin, r.expanded, r.tier) == (ll.OK, o["cost"], o["fmin"], o["expanded"], 2)
            assert [s[1:] for s in r.states] == [s[1:] for s in o["states"]]
            # the hint: every search starts with the heavy workgroups
            hinted = []
            for name, inst, c in cases[:120]:
                a = c["agent"]
                hinted.append(ll.LLJob(map_id=map_ids[name], algo=ll.ASTAR_EPS, start=inst["starts"][a], goal=inst["goals"][a],
                                       agent_idx=a, w=1.3, vertex_constraints=c["vertex_constraints"],
                                       edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"], heavy=True))
            for (name, inst, c), r in zip(cases[:120], eng.search_batch(hinted)):
                assert (r.success, r.expanded, r.tier) == (c["success"], c["expanded"], 2), name
                if c["success"]:
                    assert (r.cost, r.fmin, [s[1:] for s in r.states]) == (c["cost"], c["fmin"], c["states"])
        finally:
            eng.session_end()
        st = eng.stats()
        assert st["heavy_active_wgs"] >= 1 and st["heavy_fallbacks"] == 0
    finally:
        eng.close()
    # a horizon of 200 time steps gives the wide tier three chunks of 64 rows: the serpentine (198 steps) leaves it too and
    # ends in the arena tier
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64, max_horizon=200)
    try:
        snake_id = eng.upload_map(32, 32, obst)
        eng.session_begin_tiers(32, 4)
        try:
            r = eng.search_batch([ll.LLJob(map_id=snake_id, algo=ll.ASTAR_EPS, start=snake["starts"][0], goal=snake["goals"][0],
                                           w=1.3)])[0]
            assert (r.status, r.cost, r.fmin, r.expanded, r.tier) == (ll.OK, o["cost"], o["fmin"], o["expanded"], 1)
            assert [s[1:] for s in r.states] == [s[1:] for s in o["states"]]
        finally:
            eng.session_end()
    finally:
        eng.close()


def test_small_maps_uploaded_during_a_session(oracle_mod, bench_instances):
    """Maps smaller than a cache line uploaded while the resident kernel runs, each right after its predecessor has been
    searched: every bitmap has its own 128-byte line in the device buffer, so no XCD's L2 can serve a stale copy of the
    line a new map lands in (the kernels read obstacle words with plain cached loads)."""
    from libmultirobotplanning_amd import ll
    names = ["map_8by8_obst12_agents5_ex%d" % k for k in range(6)] + ["map_8by8_obst12_agents8_ex%d" % k for k in range(4)]
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
    try:
        first = bench_instances[names[0]]
        map_ids = {names[0]: eng.upload_map(first["dimx"], first["dimy"], first["obstacles"])}
        per_map = {name: _harvest(oracle_mod, bench_instances, [name], oracle_mod.ECBS, 1.3, 100000) for name in names}
        eng.session_begin(128)
        try:
            for rep in range(3):  # the searches of a map run on many workgroups (every XCD reads its line) before the next upload
                for name in names:
                    _run_and_compare(eng, per_map[name] * 8, ll.ASTAR_EPS, 1.3, map_ids)
        finally:
            eng.session_end()
    finally:
        eng.close()


def test_edge_cases(engine, oracle_mod):
    from libmultirobotplanning_amd import ll
    open_map = dict(dimx=4, dimy=3, obstacles=[])
    boxed = dict(dimx=3, dimy=3, obstacles=[[0, 1], [1, 0], [1, 2], [2, 1]])  # centre cell walled in
    mid_open = engine.upload_map(4, 3, [])
    mid_boxed = engine.upload_map(3, 3, boxed["obstacles"])
    specs = [
        # (map, map_id, algo, start, goal, vc, ec, ctx, w, cap)
        (open_map, mid_open, ll.ASTAR, [0, 0], [0, 0], [], [], [], 1.0, -1),                 # start == goal
        (open_map, mid_open, ll.ASTAR_EPS, [0, 0], [0, 0], [[3, 0, 0]], [], [], 1.3, -1),    # goal constrained later
        (open_map, mid_open, ll.ASTAR, [0, 0], [3, 2], [[1, 1, 0], [1, 0, 1]], [], [], 1.0, -1),  # both moves blocked at t=1
        (open_map, mid_open, ll.ASTAR_EPS, [0, 0], [3, 0], [], [[0, 0, 0, 1, 0], [1, 0, 0, 1, 0]], [], 1.3, -1),
        (boxed, mid_boxed, ll.ASTAR, [1, 1], [0, 0], [[1, 1, 1]], [], [], 1.0, -1),          # open list exhausted
        (boxed, mid_boxed, ll.ASTAR_EPS, [1, 1], [0, 0], [[1, 1, 1]], [], [], 1.3, -1),
        (boxed, mid_boxed, ll.ASTAR, [1, 1], [0, 0], [], [], [], 1.0, 50),                    # unreachable: cap hit
        (open_map, mid_open, ll.ASTAR_EPS, [0, 0], [3, 2], [], [], [[], [[1, 0], [0, 0], [0, 1]], [[3, 2]]], 1.3, -1),
    ]
    jobs = [ll.LLJob(map_id=mid, algo=algo, start=s, goal=g, agent_idx=0, w=w, vertex_constraints=vc,
                     edge_constraints=ec, ctx_paths=ctx, max_expansions=cap)
            for (_, mid, algo, s, g, vc, ec, ctx, w, cap) in specs]
    res = engine.search_batch(jobs)
    for (mp, mid, algo, s, g, vc, ec, ctx, w, cap), r in zip(specs, res):
        o = oracle_mod.ll_search(algo, mp, 0, s, g, vc, ec, ctx, w=w, cap_expansions=cap)
        if o["rc"] == -1:
            assert r.status == ll.CAP_EXPANSIONS
            continue
        assert r.success == o["success"], (s, g, vc)
        assert r.expanded == o["expanded"]
        if o["success"]:
            assert (r.cost, r.fmin, r.states, r.actions) == (o["cost"], o["fmin"], o["states"], o["actions"])
        else:
            assert r.status == ll.NO_SOLUTION
    assert engine.search_batch([]) == []
    # rejected on the host, loudly
    bad = engine.search_batch([ll.LLJob(map_id=999, algo=ll.ASTAR, start=[0, 0], goal=[1, 1])])
    assert bad[0].status == ll.BAD_JOB


def test_sipp_reference_case_and_random_jobs(engine, oracle_mod, ref_tests):
    """MRP_LL_SIPP (sipp.hpp) against the oracle: test/sipp_1.yaml (test/test_sipp.py:16-21) and random single-agent
    jobs with random collision intervals on a 16x16 grid (waits, blocked goals, several intervals per cell)."""
    import random
    from libmultirobotplanning_amd import ll
    s = ref_tests["sipp_1"]
    mid = engine.upload_map(s["dimx"], s["dimy"], s["obstacles"])
    r = engine.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=s["start"], goal=s["goal"],
                                      collision_intervals=s["collision_intervals"])])[0]
    assert r.success and len(r.states) == s["n_states"]
    assert [r.states[-1][1], r.states[-1][2], r.states[-1][0]] == s["last"]
    o_states, _ = oracle_mod.sipp_single(s["dimx"], s["dimy"], s["obstacles"], s["start"], s["goal"],
                                         s["collision_intervals"])
    assert [[x, y, t] for t, x, y in r.states] == o_states
    assert ll.ACTION_NAMES[r.actions[3]] == "Wait" and r.action_costs[3] == 5

    rng = random.Random(7)
    dim = 16
    obst = [[rng.randrange(dim), rng.randrange(dim)] for _ in range(40)]
    obst = [list(c) for c in {tuple(c) for c in obst}]
    mid2 = engine.upload_map(dim, dim, obst)
    free = [[x, y] for x in range(dim) for y in range(dim) if [x, y] not in obst]
    jobs, specs = [], []
    for case in range(150):
        st, go = rng.choice(free), rng.choice(free)
        cis = []
        for _ in range(rng.randrange(0, 60)):
            c = rng.choice(free)
            t0 = rng.randrange(0, 40)
            n_iv = rng.randrange(1, 4)
            t = t0
            for _ in range(n_iv):                      # non-overlapping intervals of one location, in order
                a = t + rng.randrange(0, 6)
                b = a + rng.randrange(0, 5)
                if rng.random() < 0.03:
                    b = 2 ** 31 - 1
                cis.append([c[0], c[1], a, b])
                t = b + 2
                if b == 2 ** 31 - 1:
                    break
        # one list per location, as the reference's setCollisionIntervals receives it
        cis.sort(key=lambda v: (v[0], v[1]))
        merged = []
        for v in cis:
            if merged and merged[-1][0][:2] == v[:2]:
                if all(v[2] > w[3] for w in merged[-1]):
                    merged[-1].append(v)
            else:
                merged.append([v])
        cis = [v for grp in merged for v in grp]
        specs.append((st, go, cis))
        jobs.append(ll.LLJob(map_id=mid2, algo=ll.SIPP, start=st, goal=go, collision_intervals=cis,
                             max_expansions=20000))
    res = engine.search_batch(jobs)
    n_ok = 0
    for (st, go, cis), r in zip(specs, res):
        o_states, o_exp = oracle_mod.sipp_single(dim, dim, obst, st, go, cis)
        if r.status == ll.CAP_EXPANSIONS:
            continue
        assert r.success == (len(o_states) > 0), (st, go)
        if r.success:
            n_ok += 1
            assert [[x, y, t] for t, x, y in r.states] == o_states, (st, go, cis)
            assert r.expanded == o_exp
            assert r.cost == o_states[-1][2]
    assert n_ok > 60
    # the same jobs through a SIPP session (resident SIPP kernel fed through the job ring): identical results; a session
    # serves one kind of job, the other kind is rejected, not guessed
    engine.session_begin_sipp(32)
    try:
        res_s = engine.search_batch(jobs + [ll.LLJob(map_id=mid2, algo=ll.ASTAR, start=free[0], goal=free[1])])
    finally:
        engine.session_end()
    assert res_s[-1].status == ll.BAD_JOB
    for a, b in zip(res, res_s[:-1]):
        assert (a.status, a.cost, a.expanded, a.states, a.actions, a.action_costs) == (
            b.status, b.cost, b.expanded, b.states, b.actions, b.action_costs)
    engine.session_begin(32)
    try:
        assert engine.search_batch(jobs[:3])[0].status == ll.BAD_JOB
    finally:
        engine.session_end()


def test_stats_report_kernel_time(engine):
    st = engine.stats()
    assert st["launches"] > 0 and st["kernel_ms"] > 0 and st["expansions"] > 0


def test_initial_cost_and_start_time(engine, oracle_mod, ref_tests):
    """The fork's extra search arguments: AStar::search(..., initialCost) (a_star.hpp:63-64,78,100) and
    SIPP::search(..., startTime) (sipp.hpp:92-103), against the oracle's restatement of both."""
    import random
    from libmultirobotplanning_amd import ll
    rng = random.Random(11)
    dim = 12
    obst = [list(c) for c in {(rng.randrange(dim), rng.randrange(dim)) for _ in range(25)}]
    m = dict(dimx=dim, dimy=dim, obstacles=obst)
    mid = engine.upload_map(dim, dim, obst)
    free = [[x, y] for x in range(dim) for y in range(dim) if [x, y] not in obst]
    jobs, specs = [], []
    for case in range(60):
        st = rng.choice(free)
        go = st if case % 10 == 0 else rng.choice(free)   # a start that already is the goal keeps fmin = h(start)
        c0 = rng.choice([0, 1, 7, 100])
        vcs = [[rng.randrange(0, 12), *rng.choice(free)] for _ in range(rng.randrange(0, 6))]
        specs.append((st, go, c0, vcs))
        jobs.append(ll.LLJob(map_id=mid, algo=ll.ASTAR, start=st, goal=go, vertex_constraints=vcs, initial_cost=c0,
                             max_expansions=20000))
    res = engine.search_batch(jobs)
    n_ok = 0
    for (st, go, c0, vcs), r in zip(specs, res):
        o = oracle_mod.ll_search(oracle_mod.ASTAR, m, 0, st, go, vertex_constraints=vcs, initial_cost=c0,
                                 cap_expansions=20000)
        if o["rc"] == -1:
            assert r.status == ll.CAP_EXPANSIONS
            continue
        assert r.success == o["success"] and r.expanded == o["expanded"], (st, go, c0)
        if r.success:
            n_ok += 1
            assert (r.cost, r.fmin) == (o["cost"], o["fmin"]), (st, go, c0)
            assert r.states == o["states"]
    assert n_ok > 30
    # A*-epsilon has no such argument: rejected, not ignored
    bad = engine.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=free[0], goal=free[1], w=1.3, initial_cost=3)])
    assert bad[0].status == ll.BAD_JOB

    # SIPP start times
    jobs, specs = [], []
    for case in range(80):
        st, go = rng.choice(free), rng.choice(free)
        cis = []
        for c in rng.sample(free, rng.randrange(0, 25)):
            t = rng.randrange(0, 30)
            for _ in range(rng.randrange(1, 3)):
                a = t + rng.randrange(0, 6)
                b = a + rng.randrange(0, 5)
                cis.append([c[0], c[1], a, b])
                t = b + 2
        t0 = rng.choice([0, 0, 3, 11, 40])
        specs.append((st, go, cis, t0))
        jobs.append(ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, collision_intervals=cis, initial_cost=t0,
                             max_expansions=20000))
    res = engine.search_batch(jobs)
    n_ok = 0
    for (st, go, cis, t0), r in zip(specs, res):
        o_states, o_exp, o_cost, o_fmin = oracle_mod.sipp_single_at(dim, dim, obst, st, go, cis, start_time=t0)
        if r.status == ll.CAP_EXPANSIONS:
            continue
        assert r.success == (len(o_states) > 0), (st, go, t0)
        if r.success:
            n_ok += 1
            assert [[x, y, t] for t, x, y in r.states] == o_states, (st, go, cis, t0)
            assert (r.expanded, r.cost, r.fmin) == (o_exp, o_cost, o_fmin), (st, go, t0)
    assert n_ok > 30


def test_specialised_sessions_and_liveness(oracle_mod, bench_instances):
    """mrp_ll_session_begin_algo: the per-algorithm resident kernels give the mixed kernel's results and reject the
    other kind; a caller that keeps polling may pause between submits longer than the idle limit; a caller that goes
    silent loses the resident kernel and is TOLD so (MRP_LL_E_DEVICE) instead of spinning."""
    import ctypes
    import os
    import time
    from libmultirobotplanning_amd import ll
    name = "map_32by32_obst204_agents10_ex1"
    inst = bench_instances[name]
    _, calls = oracle_mod.mapf_record(oracle_mod.ECBS, inst, w=1.3)
    _, calls_cbs = oracle_mod.mapf_record(oracle_mod.CBS, bench_instances["map_8by8_obst12_agents4_ex0"])
    inst8 = bench_instances["map_8by8_obst12_agents4_ex0"]
    os.environ["MRP_LL_IDLE_LIMIT_S"] = "1"
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
    try:
        mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        mid8 = eng.upload_map(inst8["dimx"], inst8["dimy"], inst8["obstacles"])
        eps_jobs = [ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                             agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]) for c in calls]
        cbs_jobs = [ll.LLJob(map_id=mid8, algo=ll.ASTAR, start=inst8["starts"][c["agent"]], goal=inst8["goals"][c["agent"]],
                             agent_idx=c["agent"], vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"]) for c in calls_cbs]

        def check(res, cs):
            for c, r in zip(cs, res):
                assert (r.success, r.expanded) == (c["success"], c["expanded"])
                if r.success:
                    assert (r.cost, r.fmin, [s[1:] for s in r.states]) == (c["cost"], c["fmin"], c["states"])

        eng.session_begin_algo(ll.ASTAR_EPS, 32)
        try:
            res = eng.search_batch(eps_jobs + cbs_jobs[:1])
            assert res[-1].status == ll.BAD_JOB
            check(res[:-1], calls)
            # pause longer than the idle limit, but keep polling: the session stays alive
            tk = (ctypes.c_int32 * 8)()
            n = ctypes.c_int32(0)
            t0 = time.time()
            while time.time() - t0 < 2.5:
                assert eng._lib.mrp_ll_poll_any(eng._h, tk, 8, ctypes.byref(n)) == 0
                time.sleep(0.05)
            check(eng.search_batch(eps_jobs), calls)
        finally:
            eng.session_end()
        eng.session_begin_algo(ll.ASTAR, 32)
        try:
            res = eng.search_batch(cbs_jobs + eps_jobs[:1])
            assert res[-1].status == ll.BAD_JOB
            check(res[:-1], calls_cbs)
            # go silent for longer than the idle limit: the resident kernel leaves, and the next wait says so
            time.sleep(3.5)
            with pytest.raises(RuntimeError):
                eng.search_batch(cbs_jobs)
        finally:
            eng.session_end()
        # the context is usable again afterwards
        eng.session_begin_algo(ll.ASTAR, 32)
        try:
            check(eng.search_batch(cbs_jobs), calls_cbs)
        finally:
            eng.session_end()
    finally:
        del os.environ["MRP_LL_IDLE_LIMIT_S"]
        eng.close()


def test_device_path_store(oracle_mod, bench_instances):
    """a search leaves its path in the engine's device-resident path store (result_path_id) and later
    A*-epsilon jobs name their focal context by slot (path_ids) instead of shipping it.  The conflict tree of a 30-agent
    instance is replayed that way — every low-level call of the oracle's run, contexts by slot — in batch mode and through
    a session; results must equal the oracle's and those of the table form."""
    from libmultirobotplanning_amd import ll
    name = "map_32by32_obst204_agents30_ex1"
    inst = bench_instances[name]
    _, calls = oracle_mod.mapf_record(oracle_mod.ECBS, inst, w=1.3)
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
    try:
        eng.path_store_reserve(4096)
        mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        slot_of = {}          # path (as tuple) -> slot where some search left it
        next_slot = [0]

        def run(session):
            slot_of.clear()
            next_slot[0] = 0
            for c in calls:
                ids = []
                ok = True
                for a, p in enumerate(c["ctx_paths"]):
                    if a == c["agent"] or not p:
                        ids.append(-1)
                    else:
                        sid = slot_of.get(tuple(map(tuple, p)))
                        ok = ok and sid is not None
                        ids.append(-1 if sid is None else sid)
                assert ok, "every context path was produced by an earlier search of this run"
                out = next_slot[0]
                next_slot[0] += 1
                job = ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                               agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                               edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"], path_ids=ids,
                               result_path_id=out)
                r = eng.search_batch([job])[0]
                assert (r.success, r.expanded) == (c["success"], c["expanded"]), (session, c["agent"])
                if r.success:
                    assert (r.cost, r.fmin, [s[1:] for s in r.states]) == (c["cost"], c["fmin"], c["states"])
                    slot_of[tuple(map(tuple, c["states"]))] = out

        run(False)
        st0 = eng.stats()["staged_bytes"]
        eng.session_begin_algo(ll.ASTAR_EPS, 16)
        try:
            run(True)
            st1 = eng.stats()["staged_bytes"]
            # the same jobs with their tables shipped: far more bytes through pinned host memory
            jobs = [ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                             agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]) for c in calls]
            res = eng.search_batch(jobs)
            st2 = eng.stats()["staged_bytes"]
        finally:
            eng.session_end()
        for c, r in zip(calls, res):
            assert (r.success, r.expanded, r.cost if r.success else 0) == (c["success"], c["expanded"], c["cost"] if c["success"] else 0)
        assert (st2 - st1) > 4 * (st1 - st0) > 0
        # a slot that does not exist is rejected on the host, loudly
        bad = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][0], goal=inst["goals"][0],
                                         w=1.3, result_path_id=999999)])
        assert bad[0].status == ll.BAD_JOB
    finally:
        eng.close()


@pytest.mark.timeout(900)
def test_sipp_device_resident_tables(oracle_mod):
    """mrp_ll_sipp_table_* inside a SIPP session: the table lives on the device and a job carries only the cells that
    changed since the table's previous job.  A prioritized-planner-like sequence (search, add intervals, search ...) on
    several tables at once must match the oracle job by job — including a table that outgrows the resident layout
    (a cell with more than 15 safe intervals: falls back to whole tables), more jobs on one table than there are epochs
    (the status words are re-zeroed), two jobs on one table in flight (the second travels whole), a start time inside a
    collision interval (no start interval; the delta must still be applied) and the table surviving session boundaries."""
    import random
    from libmultirobotplanning_amd import ll
    rng = random.Random(99)
    dim = 64
    obst = [list(c) for c in {(rng.randrange(dim), rng.randrange(dim)) for _ in range(400)}]
    oset = {tuple(c) for c in obst}
    free = [[x, y] for x in range(dim) for y in range(dim) if (x, y) not in oset]
    eng = ll.LowLevelEngine(device=0, max_cells=dim * dim)
    mid = eng.upload_map(dim, dim, obst)

    class Tab:
        def __init__(self):
            self.h = eng.sipp_table_create(mid)
            self.cis = {}      # cell -> [[a, b], ...] in time order
            self.next_t = {}

        def add(self, c, a=None, n=1):
            for _ in range(n):
                if n == 1 and rng.random() < 0.3:         # somewhere before / between the cell's earlier intervals
                    a0 = rng.randrange(0, 60)
                    b0 = a0 + rng.randrange(0, 3)
                    if any(a0 <= q[1] and q[0] <= b0 for q in self.cis.get(tuple(c), [])):
                        continue                          # the reference asserts that intervals do not overlap
                else:
                    t = self.next_t.get(tuple(c), rng.randrange(0, 30))
                    a0 = t + rng.randrange(0, 5)
                    b0 = a0 + rng.randrange(0, 4)
                self.cis.setdefault(tuple(c), []).append([a0, b0])
                self.next_t[tuple(c)] = max(self.next_t.get(tuple(c), 0), b0 + 2 + rng.randrange(0, 3))
                eng.sipp_table_add(self.h, c[0], c[1], a0, b0)

        def flat(self):
            return [[c[0], c[1], a, b] for c, v in sorted(self.cis.items()) for a, b in sorted(v)]

    def check(tab, st, go, t0, r):
        o_states, o_exp, o_cost, o_fmin = oracle_mod.sipp_single_at(dim, dim, obst, st, go, tab.flat(), start_time=t0)
        assert r.status != ll.BAD_JOB
        if r.status == ll.CAP_EXPANSIONS:
            return 0
        assert r.success == (len(o_states) > 0), (st, go, t0)
        if not r.success:
            return 0
        assert [[x, y, t] for t, x, y in r.states] == o_states, (st, go, t0)
        assert (r.expanded, r.cost, r.fmin) == (o_exp, o_cost, o_fmin)
        return 1

    tabs = [Tab() for _ in range(6)]
    n_ok = 0
    st0 = eng.stats()["staged_bytes"]
    n_jobs = 0
    eng.session_begin_sipp(32)
    try:
        for rnd in range(40):
            jobs, specs = [], []
            for k, tb in enumerate(tabs):
                st, go = rng.choice(free), rng.choice(free)
                t0 = rng.choice([0, 0, 0, 2, 9])
                if rnd == 7 and k == 1:                       # start time inside a collision interval of the start cell
                    tb.cis.setdefault(tuple(st), [])
                    a0 = tb.next_t.get(tuple(st), 0)
                    tb.cis[tuple(st)].append([a0, a0 + 3])
                    tb.next_t[tuple(st)] = a0 + 6
                    eng.sipp_table_add(tb.h, st[0], st[1], a0, a0 + 3)
                    t0 = a0 + 1
                specs.append((tb, st, go, t0))
                jobs.append(ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, sipp_table=tb.h, initial_cost=t0,
                                     max_expansions=30000))
            res = eng.search_batch(jobs)
            n_jobs += len(jobs)
            for (tb, st, go, t0), r in zip(specs, res):
                n_ok += check(tb, st, go, t0, r)
                # what a prioritized planner does next: the cells of this agent's path become collision intervals
                for c in rng.sample(free, rng.randrange(1, 25)):
                    tb.add(c)
            if rnd == 10:
                tabs[5].add(free[7], n=14)                    # 14 collision intervals -> 15 safe intervals: the layout's last slot
            if rnd == 12:                                     # a bound that does not fit the layout's halfwords: travels whole
                tabs[4].cis.setdefault(tuple(free[9]), []).append([70000, 70010])
                eng.sipp_table_add(tabs[4].h, free[9][0], free[9][1], 70000, 70010)
            if rnd == 20:
                tabs[2].add(free[5], n=20)                    # 20 collision intervals -> 21 safe intervals: beyond the layout
        small = (eng.stats()["staged_bytes"] - st0) / n_jobs
        # two jobs on ONE table in the same submit: the second cannot share the device copy
        tb = tabs[0]
        sg = [(rng.choice(free), rng.choice(free)) for _ in range(2)]
        res = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=s, goal=g_, sipp_table=tb.h) for s, g_ in sg])
        for (s, g_), r in zip(sg, res):
            n_ok += check(tb, s, g_, 0, r)
        # more jobs on one table than there are epochs (255)
        tb = tabs[3]
        for k in range(300):
            st, go = rng.choice(free), rng.choice(free)
            r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, sipp_table=tb.h,
                                           max_expansions=30000)])[0]
            if k % 10 == 0 or k in (253, 254, 255, 256):
                n_ok += check(tb, st, go, 0, r)
            if k % 3 == 0:
                tb.add(rng.choice(free))
    finally:
        eng.session_end()
    assert n_ok > 150
    # a delta is a few hundred bytes; the whole table of a 64x64 map is more than 8 KB
    assert small < 4000, small
    # outside a session the tables travel whole; back in a session the device copies pick up what changed meanwhile
    tb = tabs[4]
    for c in rng.sample(free, 10):
        tb.add(c)
    st, go = rng.choice(free), rng.choice(free)
    r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, sipp_table=tb.h)])[0]
    assert check(tb, st, go, 0, r) in (0, 1)
    tb.add(rng.choice(free))
    eng.session_begin_sipp(16)
    try:
        for tb in tabs:
            st, go = rng.choice(free), rng.choice(free)
            r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, sipp_table=tb.h)])[0]
            check(tb, st, go, 0, r)
    finally:
        eng.session_end()
    for tb in tabs:
        eng.sipp_table_destroy(tb.h)
    eng.close()


@pytest.mark.timeout(900)
def test_sipp_commit_paths_into_tables(oracle_mod):
    """mrp_ll_job.sipp_commit: the engine itself turns the path it found into collision intervals of the table (what
    mapf_prioritized_sipp.cpp:237-246 does with every solution) — in a session on the device-resident copy, by the
    workgroup that ran the search.  Several tables are planned like prioritized-planning instances, agent after agent,
    each job checked against the oracle run on the intervals the test derives from the returned paths; mixed with
    caller-made mrp_ll_sipp_table_add calls, with a commit that cannot fit the device layout (the host takes over), with
    batch mode (the host commits) and across session boundaries."""
    import random
    from libmultirobotplanning_amd import ll
    rng = random.Random(4242)
    dim = 64
    obst = [list(c) for c in {(rng.randrange(dim), rng.randrange(dim)) for _ in range(410)}]
    oset = {tuple(c) for c in obst}
    free = [[x, y] for x in range(dim) for y in range(dim) if (x, y) not in oset]
    eng = ll.LowLevelEngine(device=0, max_cells=dim * dim)
    mid = eng.upload_map(dim, dim, obst)

    class Inst:
        def __init__(self):
            self.h = eng.sipp_table_create(mid)
            self.cis = {}

        def flat(self):
            return [[c[0], c[1], a, b] for c, v in sorted(self.cis.items()) for a, b in sorted(v)]

        def free_at(self, c, a, b):
            return not any(a <= q[1] and q[0] <= b for q in self.cis.get(tuple(c), []))

        def manual(self, c, a, b):
            if self.free_at(c, a, b):
                self.cis.setdefault(tuple(c), []).append([a, b])
                eng.sipp_table_add(self.h, c[0], c[1], a, b)

        def took(self, states):                       # the stays of a returned path: one per maximal stay on a cell
            k = 0
            while k < len(states):
                j = k
                while j + 1 < len(states) and states[j + 1][1:] == states[k][1:]:
                    j += 1
                end = states[j + 1][0] - 1 if j + 1 < len(states) else 2 ** 31 - 1
                self.cis.setdefault((states[k][1], states[k][2]), []).append([states[k][0], end])
                k = j + 1

    def run_round(insts, t0s=None):
        jobs, specs = [], []
        for n, it in enumerate(insts):
            st, go = rng.choice(free), rng.choice(free)
            t0 = t0s[n] if t0s else 0
            specs.append((it, st, go, t0))
            jobs.append(ll.LLJob(map_id=mid, algo=ll.SIPP, start=st, goal=go, sipp_table=it.h, sipp_commit=True,
                                 initial_cost=t0, max_expansions=200000))
        res = eng.search_batch(jobs)
        ok = 0
        for (it, st, go, t0), r in zip(specs, res):
            o_states, o_exp, o_cost, o_fmin = oracle_mod.sipp_single_at(dim, dim, obst, st, go, it.flat(), start_time=t0)
            assert r.status in (ll.OK, ll.NO_SOLUTION), r.status
            assert r.success == (len(o_states) > 0), (st, go)
            if r.success:
                assert [[x, y, t] for t, x, y in r.states] == o_states, (st, go)
                assert (r.expanded, r.cost, r.fmin) == (o_exp, o_cost, o_fmin)
                it.took(r.states)
                ok += 1
        return ok

    insts = [Inst() for _ in range(12)]
    n_ok = 0
    eng.session_begin_sipp(32)
    try:
        for agent in range(40):
            n_ok += run_round(insts)
            if agent % 5 == 2:                        # the caller's own intervals in between
                for it in insts[:4]:
                    for _ in range(6):
                        a = rng.randrange(0, 90)
                        it.manual(rng.choice(free), a, a + rng.randrange(0, 3))
        # a stay that needs a 17th interval: cell X has sixteen safe intervals, the agent starts there at t = 2 and leaves
        it = insts[5]
        X = next(c for c in free if tuple(c) not in it.cis and all(
            0 <= c[0] + dx < dim and 0 <= c[1] + dy < dim and (c[0] + dx, c[1] + dy) not in oset and
            (c[0] + dx, c[1] + dy) not in it.cis for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1))))
        for q in range(15):
            it.manual(X, 21 + 10 * q, 21 + 10 * q)
        go = rng.choice(free)
        r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=X, goal=go, sipp_table=it.h, sipp_commit=True,
                                       initial_cost=2)])[0]
        o_states, o_exp, o_cost, o_fmin = oracle_mod.sipp_single_at(dim, dim, obst, X, go, it.flat(), start_time=2)
        assert r.success and [[x, y, t] for t, x, y in r.states] == o_states
        it.took(r.states)
        assert len([q for q in it.cis[tuple(X)]]) == 16
        for agent in range(4):                        # that table now travels whole and the host commits; the others go on
            n_ok += run_round(insts)
    finally:
        eng.session_end()
    # batch mode: no device-resident copies, the host commits
    for agent in range(3):
        n_ok += run_round(insts)
    eng.session_begin_sipp(16)
    try:
        for agent in range(3):
            n_ok += run_round(insts, t0s=[rng.choice([0, 0, 4]) for _ in insts])
    finally:
        eng.session_end()
    assert n_ok > 12 * 40
    for it in insts:
        eng.sipp_table_destroy(it.h)
    eng.close()


def test_golden_low_level_jobs(engine, bench_instances, ll_jobs_golden):
    """~200 committed low-level calls (inputs and the oracle's outputs, tests/golden/ll_jobs.json): low-level parity that
    does not need the oracle to be rebuilt on the GPU box."""
    from libmultirobotplanning_amd import ll
    assert len(ll_jobs_golden) >= 200
    for algo_name, algo_ll in (("ecbs", ll.ASTAR_EPS), ("cbs", ll.ASTAR)):
        for w in sorted({j["w"] for j in ll_jobs_golden if j["algo"] == algo_name}):
            cases = [(j["instance"], bench_instances[j["instance"]], j) for j in ll_jobs_golden
                     if j["algo"] == algo_name and j["w"] == w]
            _run_and_compare(engine, cases, algo_ll, w)


def test_zero_initialised_job_struct(oracle_mod, bench_instances):
    """A job that was memset to zero (the adapter of INTEGRATION.md: `mrp_ll_job job{}`) is a plain search: it stores its
    path nowhere — with and without a reserved path store."""
    import ctypes
    import numpy as np
    from libmultirobotplanning_amd import ll
    inst = bench_instances["map_32by32_obst204_agents10_ex1"]
    want = oracle_mod.ll_search(oracle_mod.ASTAR_EPS, dict(dimx=32, dimy=32, obstacles=inst["obstacles"]), 0,
                                inst["starts"][0], inst["goals"][0], w=1.3)
    lib = ll.load_library()
    for reserve in (0, 64):
        eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=8)
        try:
            mid = eng.upload_map(32, 32, inst["obstacles"])
            if reserve:
                assert lib.mrp_ll_path_store_reserve(eng._h, reserve) == 0
            job = ll.mrp_ll_job()
            ctypes.memset(ctypes.byref(job), 0, ctypes.sizeof(job))
            job.map_id, job.algo, job.w = mid, ll.ASTAR_EPS, 1.3
            job.start_x, job.start_y = inst["starts"][0]
            job.goal_x, job.goal_y = inst["goals"][0]
            job.max_expansions = -1
            res = ll.mrp_ll_result()
            ctypes.memset(ctypes.byref(res), 0, ctypes.sizeof(res))
            states = np.zeros((256, 3), dtype=np.int32)
            res.states_txy = states.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            res.states_cap = 256
            assert lib.mrp_ll_search_batch(eng._h, 1, ctypes.byref(job), ctypes.byref(res)) == 0
            assert (res.status, res.cost, res.expanded) == (ll.OK, want["cost"], want["expanded"])
        finally:
            eng.close()


@pytest.mark.gpu
def test_session_occupancy_depends_on_the_kernel_family():
    """mrp_ll_session_occupancy: what the runtime grants a session's resident kernel.  The A*-epsilon-only kernels keep the
    compact tier's bitmap in device memory, so their LDS window is 8 KB smaller than the CBS / mixed kernels' and more of
    their searches fit a CU; a larger path-table allowance lowers both."""
    from libmultirobotplanning_amd import ll
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
    try:
        eng.configure_tiers(2048, 64, 2048)
        eps, astar = eng.session_occupancy(ll.ASTAR_EPS), eng.session_occupancy(ll.ASTAR)
        assert 1 <= astar < eps <= 16, (eps, astar)
        eng.configure_tiers(2048, 64, 16384)
        assert eng.session_occupancy(ll.ASTAR_EPS) < eps
    finally:
        eng.close()
