"""SURVEY.md §8 f4 on the MI355X: the low level of the task-assignment callers (MRP_LL_ASTAR_TA) through the C-ABI against
the oracle's restatement of example/cbs_ta.cpp's Environment under AStar (oracle/ta_restated.hpp, pinned by the costs and end
states test/test_cbs_ta.py:24-38 asserts — tests/test_oracle_known_answers.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bfs(dimx, dimy, obstacles, goal):
    from collections import deque
    obst = {(o[0], o[1]) for o in obstacles}
    dist = [[2 ** 31 - 1] * dimx for _ in range(dimy)]
    if tuple(goal) in obst:
        return dist
    dist[goal[1]][goal[0]] = 0
    q = deque([tuple(goal)])
    while q:
        x, y = q.popleft()
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            nx, ny = x + dx, y + dy
            if 0 <= nx < dimx and 0 <= ny < dimy and (nx, ny) not in obst and dist[ny][nx] == 2 ** 31 - 1:
                dist[ny][nx] = dist[y][x] + 1
                q.append((nx, ny))
    return dist


def _run(eng, cases, ids=None):
    """cases: (map dict, start, goal or None, vc, ec, cap, oracle result).  One batch through mrp_ll_search_batch.
    ids: (maps, heuristics) already uploaded to this engine (tables cannot be uploaded during a session)."""
    from libmultirobotplanning_amd import ll
    maps, heurs = ids if ids is not None else ({}, {})
    jobs = []
    for m, s, g, vc, ec, cap, _ in cases:
        key = id(m)
        if key not in maps:
            maps[key] = eng.upload_map(m["dimx"], m["dimy"], m["obstacles"])
        hid = -1
        if g is not None:
            hk = (key, tuple(g))
            if hk not in heurs:
                heurs[hk] = eng.upload_heuristic(maps[key], _bfs(m["dimx"], m["dimy"], m["obstacles"], g))
            hid = heurs[hk]
        jobs.append(ll.LLJob(map_id=maps[key], algo=ll.ASTAR_TA, start=s, goal=g, vertex_constraints=vc, edge_constraints=ec,
                             max_expansions=cap, heuristic_id=hid))
    res = eng.search_batch(jobs)
    n = 0
    for (m, s, g, vc, ec, cap, o), r in zip(cases, res):
        # no capacity status: what the LDS tier cannot hold (64 + 64 constraints, 1023 open nodes, t <= 61, maps up to
        # 32 x 32) the arena tier runs — its own limits (mrp_ll_options.arena_nodes / max_horizon: 512 time steps on these
        # maps) are far beyond every case here
        assert r.status not in (ll.CAP_NODES, ll.CAP_HORIZON), (s, g, len(vc), len(ec), r.status)
        n += 1
        if o["rc"] == -1:
            assert r.status == ll.CAP_EXPANSIONS
            continue
        assert (r.success, r.expanded) == (o["success"], o["expanded"]), (s, g, r, o["expanded"])
        if o["success"]:
            assert r.status == ll.OK
            assert (r.cost, r.fmin, r.states, r.actions, r.action_costs) == (
                o["cost"], o["fmin"], o["states"], o["actions"], o["action_costs"]), (s, g)
        else:
            assert r.status == ll.NO_SOLUTION
    return n


def test_task_assignment_low_level_on_the_reference_fixtures(oracle_mod, ref_tests):
    """Every low-level call of cbs_ta.hpp's conflict tree over test/mapfta_simple1_a{1,2,3}.yaml, for every assignment."""
    from libmultirobotplanning_amd import ll
    from test_oracle_known_answers import _ta_assignments
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
    try:
        cases = []
        for name, inst in ref_tests["cbs_ta"]["inputs"].items():
            m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
            for tasks in _ta_assignments(inst["potential_goals"]):
                _, calls = oracle_mod.ta_cbs_fixed(m, inst["starts"], tasks)
                for c in calls:
                    o = oracle_mod.ta_ll_search(m, inst["starts"][c["agent"]], c["goal"], c["vertex_constraints"],
                                                c["edge_constraints"])
                    assert (o["success"], o["cost"], o["expanded"]) == (c["success"], c["cost"], c["expanded"])
                    cases.append((m, inst["starts"][c["agent"]], c["goal"], c["vertex_constraints"], c["edge_constraints"], -1, o))
        assert _run(eng, cases) == len(cases) >= 12
    finally:
        eng.close()


def test_task_assignment_low_level_random_constraint_sets(oracle_mod, bench_instances):
    """Random vertex / edge constraints (up to 90 of each: beyond the 64 keys a wave holds, the arena tier takes over) on
    shipped 8x8 and 32x32 maps, with and without a task, with expansion caps; batch mode and a session of A* jobs.  Not one
    capacity status (asserted in _run)."""
    from libmultirobotplanning_amd import ll
    rng = np.random.default_rng(5)
    cases = []
    maps = {}
    for trial in range(240):
        name = "map_8by8_obst12_agents8_ex%d" % (trial % 5) if trial % 2 else "map_32by32_obst204_agents10_ex%d" % (trial % 7)
        inst = bench_instances[name]
        m = maps.setdefault(name, dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"]))
        d = inst["dimx"]
        a = int(rng.integers(0, len(inst["starts"])))
        s = inst["starts"][a]
        goal = None if trial % 3 == 0 else inst["goals"][a]
        vc = [[int(rng.integers(0, 14)), int(rng.integers(0, d)), int(rng.integers(0, d))] for _ in range(int(rng.integers(0, 90)))]
        if goal is not None and trial % 4 == 1:
            vc.append([int(rng.integers(3, 20)), goal[0], goal[1]])
        ec = []
        for _ in range(int(rng.integers(0, 90))):
            x, y = int(rng.integers(0, d)), int(rng.integers(0, d))
            dx, dy = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)][int(rng.integers(0, 5))]
            ec.append([int(rng.integers(0, 14)), x, y, x + dx, y + dy])
        cap = int(rng.choice([-1, -1, -1, 25]))
        cases.append((m, s, goal, vc, ec, cap, oracle_mod.ta_ll_search(m, s, goal, vc, ec, cap_expansions=cap)))
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
    try:
        ids = ({}, {})
        assert _run(eng, cases, ids) >= 200
        eng.session_begin(64)  # (maps and heuristic tables were uploaded before the session)
        try:
            assert _run(eng, cases[:100], ids) >= 80
        finally:
            eng.session_end()
    finally:
        eng.close()


def test_task_assignment_searches_beyond_the_lds_tier(oracle_mod, bench_instances):
    """What AStar::search over cbs_ta's Environment finishes and the LDS tier cannot hold, in the arena tier (tier 1), bit
    for bit: paths far beyond 61 time steps (a 198-step serpentine; a goal constraint 150 steps out, i.e. 150 free Waits at
    the goal — decrease-key all the way), an agent without a task that must outlast a late constraint, and a 48 x 48 map
    with its [dimy][dimx] heuristic table."""
    from libmultirobotplanning_amd import ll
    obst = [[x, y] for y in range(1, 12, 2) for x in range(32) if x != (31 if (y // 2) % 2 == 0 else 0)]
    snake = dict(dimx=32, dimy=32, obstacles=obst)
    inst = bench_instances["map_32by32_obst204_agents10_ex3"]
    m32 = dict(dimx=32, dimy=32, obstacles=inst["obstacles"])
    rng = np.random.default_rng(17)
    big_obst = sorted({(int(rng.integers(0, 48)), int(rng.integers(0, 48))) for _ in range(420)})
    m48 = dict(dimx=48, dimy=48, obstacles=[list(o) for o in big_obst])
    # start and goals inside one large free component: the reference never returns from a search for an unreachable goal
    for s48 in ([x, 0] for x in range(48)):
        if tuple(s48) in set(big_obst):
            continue
        from_start = _bfs(48, 48, m48["obstacles"], s48)
        reach = sorted(((from_start[y][x], x, y) for y in range(48) for x in range(48) if from_start[y][x] < 2 ** 31 - 1), reverse=True)
        if len(reach) > 1500:
            break
    far, mid = [reach[0][1], reach[0][2]], [reach[len(reach) // 3][1], reach[len(reach) // 3][2]]
    assert reach[0][0] > 62
    cases = []
    cases.append((snake, [0, 0], [0, 12], [], [], -1))
    cases.append((snake, [0, 0], [0, 12], [[230, 0, 12]], [[100, 12, 6, 13, 6]], -1))
    g = inst["goals"][2]
    cases.append((m32, inst["starts"][2], g, [[150, g[0], g[1]]], [], -1))                     # waits at the goal until t = 151
    cases.append((m32, inst["starts"][4], None, [[120, inst["starts"][4][0], inst["starts"][4][1]], [90, 3, 3]], [], -1))
    cases.append((m48, s48, far, [], [], -1))
    cases.append((m48, s48, mid, [[70, mid[0], mid[1]], [10, s48[0], s48[1]]], [[0, s48[0], s48[1], s48[0], s48[1]]], -1))
    cases = [(m, s, gl, vc, ec, cap, oracle_mod.ta_ll_search(m, s, gl, vc, ec, cap_expansions=cap, cap=1024)) for m, s, gl, vc, ec, cap in cases]
    assert all(c[6]["success"] for c in cases) and cases[0][6]["cost"] == 198 and len(cases[2][6]["states"]) >= 152
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
    try:
        assert _run(eng, cases) == len(cases)
        maps, heurs = {}, {}
        jobs = []
        for m, s, gl, vc, ec, cap, _ in cases:
            key = id(m)
            if key not in maps:
                maps[key] = eng.upload_map(m["dimx"], m["dimy"], m["obstacles"])
            hid = -1
            if gl is not None:
                hid = eng.upload_heuristic(maps[key], _bfs(m["dimx"], m["dimy"], m["obstacles"], gl))
            jobs.append(ll.LLJob(map_id=maps[key], algo=ll.ASTAR_TA, start=s, goal=gl, vertex_constraints=vc, edge_constraints=ec,
                                 max_expansions=cap, heuristic_id=hid))
        assert [r.tier for r in eng.search_batch(jobs)] == [1] * len(jobs)
    finally:
        eng.close()
