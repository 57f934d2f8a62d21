"""The compact search tier (libmultirobotplanning_amd/csrc/ll_compact.h — the source the gfx950 kernels compile) replayed
on the CPU: the same file built against tests/support/wave_emu.h, a 64-lane lockstep interpretation of its wave vocabulary,
and run on low-level searches harvested from the oracle's CBS / ECBS conflict trees.  Every search the tier finishes must
equal the oracle bit for bit (success, cost, fmin, expansions, path); a search it hands over (overflow) is the arena
tier's business and is only counted."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I32P = ctypes.POINTER(ctypes.c_int32)
I64P = ctypes.POINTER(ctypes.c_int64)


def _emu_lib(force_walk=False):
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    groups = os.environ.get("MRP_CT_GROUPS")  # the narrow tier with another number of 256-entry groups (the product builds 4)
    sanitize = bool(os.environ.get("MRP_EMU_SANITIZE"))
    # the build flavour is part of the file name: a sanitizer build left behind can never be picked up by a plain run
    # (dlopen of an ASan library into a plain interpreter ends the process without a report)
    lib = os.path.join(build, "libemu_ll%s%s%s.so" % ("_g" + groups if groups else "", "_san" if sanitize else "",
                                                      "_fw" if force_walk else ""))
    deps = [os.path.join(ROOT, "tests", "support", "emu_ll.cpp"), os.path.join(ROOT, "tests", "support", "wave_emu.h"),
            os.path.join(ROOT, "libmultirobotplanning_amd", "csrc", "ll_compact.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if sanitize else []
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"] + san +
                              (["-DMRP_CT_NARROW_GROUPS=" + groups] if groups else []) +
                              (["-DMRP_CT_FORCE_WALK"] if force_walk else []) + ["-o", lib, deps[0]])
    L = ctypes.CDLL(lib)
    L.emu_compact_search.restype = ctypes.c_int
    L.emu_compact_search.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, I32P, ctypes.c_int, I32P,
                                     ctypes.c_int, ctypes.c_int, I32P, I32P, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, I64P, I32P, ctypes.c_int]
    return L


def _arr(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(shape))
    return a, a.ctypes.data_as(I32P)


def emu_search(L, eps, inst, agent, start, goal, vc, ec, ctx_paths, w, max_exp=-1, lds_path_bytes=2048, open_cap=0,
               max_t=0):
    obst, obst_p = _arr(inst["obstacles"], (-1, 2))
    vca, vc_p = _arr(vc, (-1, 3))
    eca, ec_p = _arr(ec, (-1, 5))
    plen, plen_p = _arr([len(p) for p in ctx_paths], (-1,))
    pxy, pxy_p = _arr([xy for p in ctx_paths for xy in p], (-1, 2))
    out = np.zeros(8, dtype=np.int64)
    states = np.zeros((1024, 2), dtype=np.int32)
    rc = L.emu_compact_search((3 if getattr(L, "_wide", False) else 2 if getattr(L, "_bg", False) else 1) if eps else 0, inst["dimx"], inst["dimy"], len(obst), obst_p, start[0], start[1], goal[0],
                              goal[1], w, len(vca), vc_p, len(eca), ec_p, len(plen), agent, plen_p, pxy_p, max_exp,
                              lds_path_bytes, open_cap, max_t, out.ctypes.data_as(I64P), states.ctypes.data_as(I32P), 1024)
    if rc == -2:  # not a job of the compact tier (more than 64 edge constraints, more than 128 agents)
        return dict(status=-1, cost=0, expanded=0, oob_reads=0, oob_writes=0)
    assert rc == 0, rc
    return dict(status=int(out[0]), cost=int(out[1]), fmin=int(out[2]), n_states=int(out[3]), expanded=int(out[4]),
                nodes=int(out[5]), oob_reads=int(out[6]), oob_writes=int(out[7]), states=states[:int(out[3])].tolist())


def _replay(L, oracle_mod, inst, algo, w, lds_path_bytes=2048, cap_total=-1, open_cap=0, max_t=0):
    """Every low-level call of the instance's conflict tree through the emulated tier.  Returns (finished, handed over)."""
    eps = algo == oracle_mod.ECBS
    _, calls = oracle_mod.mapf_record(algo, inst, w=w, cap_total=cap_total)
    done = over = 0
    for c in calls:
        r = emu_search(L, eps, inst, c["agent"], inst["starts"][c["agent"]], inst["goals"][c["agent"]],
                       c["vertex_constraints"], c["edge_constraints"], c["ctx_paths"] if eps else [], w,
                       lds_path_bytes=lds_path_bytes, open_cap=open_cap, max_t=max_t)
        assert r["oob_writes"] == 0 and r["oob_reads"] == 0, r
        if r["status"] == -1:
            over += 1
            continue
        done += 1
        assert (r["status"] == 0) == c["success"], (c["agent"], r["status"])
        assert r["expanded"] == c["expanded"], (c["agent"], r["expanded"], c["expanded"])
        if c["success"]:
            assert (r["cost"], r["fmin"]) == (c["cost"], c["fmin"]), (c["agent"], r, c["cost"], c["fmin"])
            assert r["states"] == c["states"], c["agent"]
    return done, over


@pytest.fixture(scope="module", params=[0, 1, 2], ids=["bitmap_in_lds", "bitmap_in_memory", "wide"])
def emu(request):
    """The forms of the tier: the (time, cell) bitmap in the LDS window (CBS / mixed kernels), in device memory
    (ll_compact.h BG: the A*-epsilon-only kernels, whose window is 8 KB smaller), and the wide geometry of the heavy
    workgroups (3071 open entries, entries that carry h instead of g: long horizons; A*-epsilon only — its A* searches run the narrow form)."""
    L = _emu_lib()
    L._bg = request.param >= 1
    L._wide = request.param == 2
    return L


def test_ecbs_agents10_all_searches(emu, oracle_mod, bench_instances):
    done = over = 0
    for k in range(0, 100, 4):
        d, o = _replay(emu, oracle_mod, bench_instances["map_32by32_obst204_agents10_ex%d" % k], oracle_mod.ECBS, 1.3)
        done += d
        over += o
    assert done > 200 and over < done // 3, (done, over)


def test_skipping_walks_with_an_empty_band_is_unobservable(oracle_mod, bench_instances):
    """a_star_epsilon.hpp:141-152: the ordered walk only pushes the open nodes of the band old * w < f <= new * w into the
    focal list; the tier skips it when no open node is in the band.  Same searches through a build that always walks
    (-DMRP_CT_FORCE_WALK): every output word equal."""
    skip, walk = _emu_lib(), _emu_lib(force_walk=True)
    n = 0
    for form in (0, 1, 2):
        skip._bg = walk._bg = form >= 1
        skip._wide = walk._wide = form == 2
        for name in ["map_32by32_obst204_agents10_ex%d" % k for k in (1, 5, 9, 13)] + ["map_32by32_obst204_agents50_ex3"]:
            inst = bench_instances[name]
            _, calls = oracle_mod.mapf_record(oracle_mod.ECBS, inst, w=1.3)
            for c in calls:
                args = (True, inst, c["agent"], inst["starts"][c["agent"]], inst["goals"][c["agent"]],
                        c["vertex_constraints"], c["edge_constraints"], c["ctx_paths"], 1.3)
                assert emu_search(skip, *args) == emu_search(walk, *args), (name, c["agent"])
                n += 1
    assert n >= 200


def test_ecbs_denser_instances_tables_in_lds_and_in_memory(emu, oracle_mod, bench_instances):
    done = over = 0
    for name, pb in (("map_32by32_obst204_agents20_ex0", 2048), ("map_32by32_obst204_agents30_ex1", 0),
                     ("map_32by32_obst204_agents50_ex3", 16384), ("map_32by32_obst204_agents50_ex5", 0),
                     ("map_32by32_obst204_agents100_ex2", 0), ("map_32by32_obst204_agents100_ex5", 16384)):
        # (agents100_ex5 has searches whose open list crosses the 256-entry scan groups with the popped node at a boundary)
        d, o = _replay(emu, oracle_mod, bench_instances[name], oracle_mod.ECBS, 1.3, lds_path_bytes=pb)
        done += d
        over += o
    assert done > 450, (done, over)
    if emu._wide:  # the searches that outgrow the narrow geometry (open lists beyond 1023 entries) finish in the wide one
        assert over == 0, over


def test_ecbs_w1_and_cbs_small_maps(emu, oracle_mod, bench_instances, ref_tests):
    done = 0
    for name in ("map_8by8_obst12_agents6_ex1", "map_8by8_obst12_agents8_ex3", "map_8by8_obst12_agents5_ex0"):
        d, _ = _replay(emu, oracle_mod, bench_instances[name], oracle_mod.CBS, 1.0)
        done += d
        d, _ = _replay(emu, oracle_mod, bench_instances[name], oracle_mod.ECBS, 1.0)
        done += d
    assert done > 50


def test_random_constraint_sets_and_caps(emu, oracle_mod, bench_instances):
    """Random vertex / edge constraints (more than a wave of edge constraints too), goal constraints, expansion caps."""
    rng = np.random.default_rng(7)
    inst = bench_instances["map_32by32_obst204_agents10_ex7"]
    m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
    n = 0
    for trial in range(60):
        a = int(rng.integers(0, 10))
        s, g = inst["starts"][a], inst["goals"][a]
        nvc, nec = int(rng.integers(0, 40)), int(rng.choice([0, 3, 20, 70, 130]))
        vc = [[int(rng.integers(0, 40)), int(rng.integers(0, 32)), int(rng.integers(0, 32))] for _ in range(nvc)]
        if trial % 3 == 0:
            vc.append([int(rng.integers(5, 30)), g[0], g[1]])  # m_lastGoalConstraint
        ec = []
        for _ in range(nec):
            x, y = int(rng.integers(0, 32)), int(rng.integers(0, 32))
            dx, dy = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)][int(rng.integers(0, 5))]
            ec.append([int(rng.integers(0, 30)), x, y, x + dx, y + dy])
        ctx = [[] for _ in range(10)]
        for b in range(10):
            if b != a and rng.random() < 0.7:
                ctx[b] = oracle_mod.ll_search(oracle_mod.ASTAR, m, b, inst["starts"][b], inst["goals"][b])["states"]
                ctx[b] = [st[1:] for st in ctx[b]]
        cap = int(rng.choice([-1, -1, 5, 40]))
        for algo, eps, w in ((oracle_mod.ASTAR_EPS, True, 1.3), (oracle_mod.ASTAR, False, 1.0)):
            o = oracle_mod.ll_search(algo, m, a, s, g, vc, ec, ctx if eps else [], w=w, cap_expansions=cap)
            r = emu_search(emu, eps, inst, a, s, g, vc, ec, ctx if eps else [], w, max_exp=cap)
            if r["status"] == -1:
                continue
            n += 1
            if o["rc"] == -1:
                assert r["status"] == 2, (trial, r)
                continue
            assert (r["status"] == 0) == o["success"] and r["expanded"] == o["expanded"], (trial, r, o["expanded"])
            if o["success"]:
                assert (r["cost"], r["fmin"]) == (o["cost"], o["fmin"])
                assert r["states"] == [st[1:] for st in o["states"]]
    assert n > 60


def test_long_horizons(emu, oracle_mod):
    """Paths far beyond 61 time steps (a 198-step serpentine, with vertex / edge / goal constraints late on the way): the
    narrow forms hand them over, the wide geometry — entries that carry h instead of g, bitmap rows built 64 at a time as t
    grows — finishes them bit for bit."""
    obst = [[x, y] for y in range(1, 12, 2) for x in range(32) if x != (31 if (y // 2) % 2 == 0 else 0)]
    m = dict(dimx=32, dimy=32, obstacles=obst)
    s, g = [0, 0], [0, 12]
    free = oracle_mod.ll_search(oracle_mod.ASTAR_EPS, m, 0, s, g, [], [], [], w=1.3)
    assert free["success"] and free["cost"] == 198
    on_path = [st[1:] for st in free["states"]]
    cases = [([], [])]
    cases.append(([[150, on_path[150][0], on_path[150][1]], [70, on_path[70][0], on_path[70][1]]], []))   # forces waits
    cases.append(([[230, g[0], g[1]]], [[100, on_path[100][0], on_path[100][1], on_path[101][0], on_path[101][1]]]))  # late goal constraint
    other = [[[31, 20]] * 5 + on_path[5:60]]  # another agent walking the same corridor: focal conflicts
    n = 0
    for vc, ec in cases:
        for ctx in ([], [[], other[0]]):
            o = oracle_mod.ll_search(oracle_mod.ASTAR_EPS, m, 0, s, g, vc, ec, ctx, w=1.3)
            r = emu_search(emu, True, dict(m, starts=[s], goals=[g]), 0, s, g, vc, ec, ctx, 1.3)
            assert r["oob_reads"] == 0 and r["oob_writes"] == 0
            if not emu._wide:
                assert r["status"] == -1  # handed over
                continue
            assert (r["status"] == 0, r["expanded"]) == (o["success"], o["expanded"]), (vc, ec, r["status"], r["expanded"], o["expanded"])
            assert (r["cost"], r["fmin"], r["states"]) == (o["cost"], o["fmin"], [st[1:] for st in o["states"]])
            n += 1
    assert n == (6 if emu._wide else 0)


def test_tight_limits_hand_over_cleanly(emu, oracle_mod, bench_instances):
    """Small open-list / time-step limits (mrp_ll_configure_tiers): what still finishes inside the tier is exact, the rest
    is handed over."""
    inst = bench_instances["map_32by32_obst204_agents10_ex9"]
    d1, o1 = _replay(emu, oracle_mod, inst, oracle_mod.ECBS, 1.3, open_cap=40)
    d2, o2 = _replay(emu, oracle_mod, inst, oracle_mod.ECBS, 1.3, max_t=12)
    d3, o3 = _replay(emu, oracle_mod, inst, oracle_mod.ECBS, 1.3, open_cap=4)
    assert o1 > 0 and o2 > 0 and d1 > 0 and d3 == 0, (d1, o1, d2, o2, d3, o3)


def test_golden_low_level_jobs_through_the_emulated_tier(emu, bench_instances, ll_jobs_golden):
    """tests/golden/ll_jobs.json (committed inputs and oracle outputs) through the emulated tier: no oracle involved."""
    n = 0
    for j in ll_jobs_golden:
        inst = bench_instances[j["instance"]]
        a = j["agent"]
        eps = j["algo"] == "ecbs"
        r = emu_search(emu, eps, inst, a, inst["starts"][a], inst["goals"][a], j["vertex_constraints"], j["edge_constraints"],
                       j["ctx_paths"] if eps else [], j["w"], lds_path_bytes=8192)
        if r["status"] == -1:
            continue
        n += 1
        assert (r["status"] == 0, r["expanded"]) == (j["success"], j["expanded"]), (j["instance"], a)
        if j["success"]:
            assert (r["cost"], r["fmin"], r["states"]) == (j["cost"], j["fmin"], j["states"]), (j["instance"], a)
    assert n >= 190


def _bfs_table(dimx, dimy, obstacles, goal):
    """shortest_path_heuristic.hpp's row for `goal` (what a caller uploads with mrp_ll_upload_heuristic)."""
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


def emu_search_ta(L, inst_map, start, goal, vc, ec, max_exp=-1, open_cap=0, max_t=0):
    if not hasattr(L, "_ta_ready"):
        L.emu_compact_search_ta.restype = ctypes.c_int
        L.emu_compact_search_ta.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, I32P, ctypes.c_int, I32P, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_int, I64P, I32P, ctypes.c_int]
        L._ta_ready = True
    obst, obst_p = _arr(inst_map["obstacles"], (-1, 2))
    vca, vc_p = _arr(vc, (-1, 3))
    eca, ec_p = _arr(ec, (-1, 5))
    g = goal if goal is not None else (0, 0)
    heur, heur_p = _arr(_bfs_table(inst_map["dimx"], inst_map["dimy"], inst_map["obstacles"], g) if goal is not None else [0], (-1,))
    out = np.zeros(8, dtype=np.int64)
    states = np.zeros((1024, 2), dtype=np.int32)
    rc = L.emu_compact_search_ta(inst_map["dimx"], inst_map["dimy"], len(obst), obst_p, start[0], start[1], 0 if goal is None else 1,
                                 g[0], g[1], heur_p, len(vca), vc_p, len(eca), ec_p, max_exp, open_cap, max_t,
                                 out.ctypes.data_as(I64P), states.ctypes.data_as(I32P), 1024)
    assert rc == 0, rc
    return dict(status=int(out[0]), cost=int(out[1]), fmin=int(out[2]), n_states=int(out[3]), expanded=int(out[4]),
                oob_reads=int(out[6]), oob_writes=int(out[7]), states=states[:int(out[3])].tolist())


def _compare_ta(r, o):
    assert r["oob_reads"] == 0 and r["oob_writes"] == 0
    if r["status"] in (3, 4):
        return False  # a capacity limit of the tier
    if o["rc"] == -1:
        assert r["status"] == 2
        return True
    assert (r["status"] == 0, r["expanded"]) == (o["success"], o["expanded"]), (r, o["expanded"])
    if o["success"]:
        assert (r["cost"], r["fmin"]) == (o["cost"], o["fmin"]), (r, o["cost"], o["fmin"])
        assert r["states"] == [s[1:] for s in o["states"]]
    return True


def test_task_assignment_low_level(emu, oracle_mod, ref_tests, bench_instances):
    if emu._bg:
        pytest.skip("the task-assignment search has one form (bitmap in LDS)")
    """SURVEY.md §8 f4: the compact tier's search for the task-assignment callers (optional goal, shortest-path heuristic,
    free Wait at the goal, decrease-key live) against the oracle's restatement of example/cbs_ta.cpp + a_star.hpp: every
    low-level call of the conflict trees over the reference's three fixtures (all assignments), then random constraint
    sets on shipped 8x8 and 32x32 maps, with and without a task."""
    from test_oracle_known_answers import _ta_assignments
    n = 0
    for name, inst in ref_tests["cbs_ta"]["inputs"].items():
        m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
        for tasks in _ta_assignments(inst["potential_goals"]):
            _, calls = oracle_mod.ta_cbs_fixed(m, inst["starts"], tasks)
            for c in calls:
                o = dict(rc=0, success=c["success"], cost=c["cost"], fmin=c["fmin"], expanded=c["expanded"], states=c["states"])
                r = emu_search_ta(emu, m, inst["starts"][c["agent"]], c["goal"], c["vertex_constraints"], c["edge_constraints"])
                n += _compare_ta(r, o)
    assert n >= 12
    rng = np.random.default_rng(11)
    for trial in range(120):
        inst = bench_instances["map_8by8_obst12_agents8_ex%d" % (trial % 5)] if trial % 2 else \
            bench_instances["map_32by32_obst204_agents10_ex%d" % (trial % 7)]
        m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
        d = inst["dimx"]
        a = int(rng.integers(0, len(inst["starts"])))
        s = inst["starts"][a]
        goal = None if trial % 3 == 0 else inst["goals"][a]
        vc = [[int(rng.integers(0, 14)), int(rng.integers(0, d)), int(rng.integers(0, d))] for _ in range(int(rng.integers(0, 30)))]
        if goal is not None and trial % 4 == 1:
            vc.append([int(rng.integers(3, 20)), goal[0], goal[1]])
        ec = []
        for _ in range(int(rng.integers(0, 30))):
            x, y = int(rng.integers(0, d)), int(rng.integers(0, d))
            dx, dy = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)][int(rng.integers(0, 5))]
            ec.append([int(rng.integers(0, 14)), x, y, x + dx, y + dy])
        cap = int(rng.choice([-1, -1, -1, 30]))
        o = oracle_mod.ta_ll_search(m, s, goal, vc, ec, cap_expansions=cap)
        r = emu_search_ta(emu, m, s, goal, vc, ec, max_exp=cap)
        n += _compare_ta(r, o)
    assert n >= 100
