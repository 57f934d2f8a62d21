"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/liboracle.so``: the CPU restatement of the reference's low-level searches
(a_star.hpp / a_star_epsilon.hpp / sipp.hpp), of the grid MAPF Environment (example/ecbs.cpp, example/cbs.cpp) and of
the CBS / ECBS conflict-tree loops (cbs.hpp / ecbs.hpp).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product (``libmultirobotplanning_amd``) never does.

Pinning status (DESIGN.md "Oracle pinning"): pinned by every known-answer assertion of the reference's own tests
(test/test_a_star.py, test_cbs.py, test_ecbs.py, test_sipp.py, test_mapf_prioritized_sipp.py).  The reference itself is
unbuildable in this image (Boost.Heap / Boost.Program_options / yaml-cpp absent), and none of its tests pins ECBS with
w > 1, so ECBS w=1.3 tie-break parity with a real Boost build is *unpinned* beyond the restated heap rules.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I32P = ctypes.POINTER(ctypes.c_int32)
I64P = ctypes.POINTER(ctypes.c_int64)
U8P = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/liboracle.so with g++ (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_mapf_solve.restype = ctypes.c_int
        _LIB.oracle_mapf_solve.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P,
                                           ctypes.c_int, I32P, I32P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           I64P, I32P, I32P, ctypes.c_int]
        _LIB.oracle_mapf_solve_batch.restype = ctypes.c_int64
        _LIB.oracle_mapf_solve_batch.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, I32P, ctypes.c_int, I32P, I32P, ctypes.c_int64,
                                                 ctypes.c_int, I64P]
        _LIB.oracle_mapf_solve_batch_digest.restype = ctypes.c_int64
        _LIB.oracle_mapf_solve_batch_digest.argtypes = _LIB.oracle_mapf_solve_batch.argtypes
        _LIB.oracle_conflict_scan.restype = None
        _LIB.oracle_conflict_scan.argtypes = [ctypes.c_int, I32P, I32P, I32P]
        _LIB.oracle_prioritized_sipp_batch.restype = ctypes.c_int64
        _LIB.oracle_prioritized_sipp_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P,
                                                       ctypes.c_int, I32P, I32P, ctypes.c_int, I64P]
        _LIB.oracle_ll_search.restype = ctypes.c_int
        _LIB.oracle_ll_search.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, I32P, ctypes.c_int, I32P, ctypes.c_int, I32P, I32P,
                                          ctypes.c_int64, I32P, I64P, I32P, I32P, ctypes.c_int]
        _LIB.oracle_ll_search_init.restype = ctypes.c_int
        _LIB.oracle_ll_search_init.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, I32P, ctypes.c_int, I32P, ctypes.c_int, I32P, I32P,
                                               ctypes.c_int64, ctypes.c_int, I32P, I64P, I32P, I32P, ctypes.c_int]
        _LIB.oracle_sipp_single_at.restype = ctypes.c_int
        _LIB.oracle_sipp_single_at.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, I32P,
                                               ctypes.c_int, I64P, I32P]
        _LIB.oracle_mapf_record.restype = ctypes.c_int64
        _LIB.oracle_mapf_record.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            I32P, ctypes.c_int, I32P, I32P, ctypes.c_int64, I32P, ctypes.c_int64,
                                            I32P, I64P]
        _LIB.oracle_heap_replay.restype = ctypes.c_int
        _LIB.oracle_heap_replay.argtypes = [ctypes.c_int, I32P, I32P, I32P, I32P]
        _LIB.oracle_astar_2d.restype = ctypes.c_int
        _LIB.oracle_astar_2d.argtypes = [ctypes.c_int, ctypes.c_int, U8P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, I32P, ctypes.c_int, I64P]
        _LIB.oracle_prioritized_sipp.restype = ctypes.c_int
        _LIB.oracle_prioritized_sipp.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, I32P,
                                                 I32P, I64P, I32P, I32P, I32P, ctypes.c_int]
        _LIB.oracle_ta_ll_search.restype = ctypes.c_int
        _LIB.oracle_ta_ll_search.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, I32P,
                                             ctypes.c_int64, I32P, I64P, I32P, I32P, I32P, ctypes.c_int]
        _LIB.oracle_ta_cbs_fixed.restype = ctypes.c_int64
        _LIB.oracle_ta_cbs_fixed.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, I32P, I32P, I64P,
                                             I32P, I32P, ctypes.c_int64, I32P]
        _LIB.oracle_sipp_single.restype = ctypes.c_int
        _LIB.oracle_sipp_single.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, I32P, ctypes.c_int, I64P]
    return _LIB


_BOOST = None


def boost_lib():
    """The same oracle built on the REAL boost::heap::d_ary_heap (oracle/boost_heap_adapter.hpp, `make liboracle_boost.so`), or
    None where Boost.Heap is not installed (this image) or the build fails."""
    global _BOOST
    if _BOOST is None:
        so = os.path.join(_HERE, "liboracle_boost.so")
        try:
            subprocess.check_call(["make", "-C", _HERE, "liboracle_boost.so"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            L = ctypes.CDLL(so)
            L.oracle_heap_kind.restype = ctypes.c_int
            if L.oracle_heap_kind() != 1:
                raise OSError("not a Boost build")
            L.oracle_mapf_solve_batch.restype = ctypes.c_int64
            L.oracle_mapf_solve_batch.argtypes = lib().oracle_mapf_solve_batch.argtypes
            _BOOST = L
        except (OSError, subprocess.CalledProcessError):
            _BOOST = False
    return _BOOST or None


def boost_crosscheck(algo, dimx, dimy, obstacles, starts, goals, per_restated, w=1.0, cap_total=-1, n_threads=1):
    """Runs the instances through the Boost-backed oracle and compares (rc, cost, makespan, highLevelExpanded,
    lowLevelExpanded) with `per_restated` (the [n][6] array mapf_solve_batch returned for the same instances).
    "absent" | "identical" | {"mismatches": k, "of": n}."""
    L = boost_lib()
    if L is None:
        return "absent"
    ob = np.ascontiguousarray(obstacles, dtype=np.int32)
    st = np.ascontiguousarray(starts, dtype=np.int32)
    go = np.ascontiguousarray(goals, dtype=np.int32)
    n = len(st)
    out = np.zeros((n, 6), dtype=np.int64)
    L.oracle_mapf_solve_batch(algo, w, n, dimx, dimy, ob.shape[1], ob.ctypes.data_as(I32P), st.shape[1], st.ctypes.data_as(I32P),
                              go.ctypes.data_as(I32P), cap_total, n_threads, out.ctypes.data_as(I64P))
    bad = int((out[:, :5] != np.asarray(per_restated)[:n, :5]).any(axis=1).sum())
    return "identical" if bad == 0 else {"mismatches": bad, "of": n}


def _i32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    return a, a.ctypes.data_as(I32P)


CBS, ECBS = 0, 1
ASTAR, ASTAR_EPS = 0, 1


def mapf_solve(algo, inst, w=1.0, cap_per_search=-1, cap_total=-1, cap_hl=-1, path_cap=512):
    """inst = dict(dimx, dimy, obstacles [[x,y]..], starts [[x,y]..], goals [[x,y]..]) -> dict of results."""
    obst, obst_p = _i32(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
    starts, starts_p = _i32(inst["starts"])
    goals, goals_p = _i32(inst["goals"])
    n = len(starts)
    stats = np.zeros(6, dtype=np.int64)
    plen = np.zeros(n, dtype=np.int32)
    paths = np.zeros((n, path_cap, 2), dtype=np.int32)
    rc = lib().oracle_mapf_solve(algo, w, inst["dimx"], inst["dimy"], len(obst), obst_p, n, starts_p, goals_p,
                                 cap_per_search, cap_total, cap_hl, stats.ctypes.data_as(I64P),
                                 plen.ctypes.data_as(I32P), paths.ctypes.data_as(I32P), path_cap)
    out = dict(rc=rc, cost=int(stats[0]), makespan=int(stats[1]), hl_expanded=int(stats[2]),
               ll_expanded=int(stats[3]), elapsed_ns=int(stats[4]))
    if rc == 1:
        assert plen.max() <= path_cap
        out["paths"] = [paths[i, :plen[i]].tolist() for i in range(n)]
    return out


def ll_search(algo, inst_map, agent_idx, start, goal, vertex_constraints=(), edge_constraints=(), ctx_paths=(),
              w=1.0, cap_expansions=-1, cap=1024, initial_cost=0):
    """One low-level search. ctx_paths: list (per agent) of [[x,y]..] (empty list = empty path)."""
    obst, obst_p = _i32(np.asarray(inst_map["obstacles"], dtype=np.int32).reshape(-1, 2))
    vc, vc_p = _i32(np.asarray(vertex_constraints, dtype=np.int32).reshape(-1, 3))
    ec, ec_p = _i32(np.asarray(edge_constraints, dtype=np.int32).reshape(-1, 5))
    ctx_len, ctx_len_p = _i32([len(p) for p in ctx_paths])
    flat = [xy for p in ctx_paths for xy in p]
    ctx_xy, ctx_xy_p = _i32(np.asarray(flat, dtype=np.int32).reshape(-1, 2))
    out = np.zeros(4, dtype=np.int32)
    expanded = np.zeros(1, dtype=np.int64)
    states = np.zeros((cap, 3), dtype=np.int32)
    actions = np.zeros(cap, dtype=np.int32)
    rc = lib().oracle_ll_search_init(algo, w, inst_map["dimx"], inst_map["dimy"], len(obst), obst_p, agent_idx,
                                     start[0], start[1], goal[0], goal[1], len(vc), vc_p, len(ec), ec_p, len(ctx_len),
                                     ctx_len_p, ctx_xy_p, cap_expansions, initial_cost, out.ctypes.data_as(I32P),
                                     expanded.ctypes.data_as(I64P), states.ctypes.data_as(I32P),
                                     actions.ctypes.data_as(I32P), cap)
    n = int(out[3])
    assert n <= cap
    return dict(rc=rc, success=bool(out[0]), cost=int(out[1]), fmin=int(out[2]), expanded=int(expanded[0]),
                states=states[:n].tolist(), actions=actions[:max(n - 1, 0)].tolist())


def mapf_record(algo, inst, w=1.0, cap_total=-1):
    """Run CBS/ECBS and return (summary, [low-level calls]) with every LL call's inputs and outputs."""
    obst, obst_p = _i32(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
    starts, starts_p = _i32(inst["starts"])
    goals, goals_p = _i32(inst["goals"])
    n = len(starts)
    ncalls = np.zeros(1, dtype=np.int32)
    stats = np.zeros(4, dtype=np.int64)
    words = 1 << 20
    while True:
        buf = np.zeros(words, dtype=np.int32)
        need = lib().oracle_mapf_record(algo, w, inst["dimx"], inst["dimy"], len(obst), obst_p, n, starts_p, goals_p,
                                        cap_total, buf.ctypes.data_as(I32P), words, ncalls.ctypes.data_as(I32P),
                                        stats.ctypes.data_as(I64P))
        if need <= words:
            break
        words = int(need)
    calls = []
    p = 0
    for _ in range(int(ncalls[0])):
        agent, success, cost, fmin, expanded, nvc, nec, nctx, nst = (int(v) for v in buf[p:p + 9])
        p += 9
        vc = buf[p:p + 3 * nvc].reshape(-1, 3).tolist(); p += 3 * nvc
        ec = buf[p:p + 5 * nec].reshape(-1, 5).tolist(); p += 5 * nec
        clen = buf[p:p + nctx].tolist(); p += nctx
        ctx = []
        for L in clen:
            ctx.append(buf[p:p + 2 * L].reshape(-1, 2).tolist()); p += 2 * L
        st = buf[p:p + 2 * nst].reshape(-1, 2).tolist(); p += 2 * nst
        calls.append(dict(agent=agent, success=bool(success), cost=cost, fmin=fmin, expanded=expanded,
                          vertex_constraints=vc, edge_constraints=ec, ctx_paths=ctx, states=st))
    summary = dict(cost=int(stats[0]), rc=int(stats[1]), hl_expanded=int(stats[2]), ll_expanded=int(stats[3]))
    return summary, calls


def heap_replay(ops):
    ops_a, ops_p = _i32(np.asarray(ops, dtype=np.int32).reshape(-1, 3))
    cap = len(ops_a) + 1
    layout = np.zeros(cap, dtype=np.int32)
    w1 = np.zeros(cap, dtype=np.int32)
    w2 = np.zeros(cap, dtype=np.int32)
    n = lib().oracle_heap_replay(len(ops_a), ops_p, layout.ctypes.data_as(I32P), w1.ctypes.data_as(I32P),
                                 w2.ctypes.data_as(I32P))
    return layout[:n].tolist(), w1[:n].tolist(), w2[:n].tolist()


def astar_2d(dimx, dimy, mask, start, goal, cap=4096):
    m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    states = np.zeros((cap, 2), dtype=np.int32)
    expanded = np.zeros(1, dtype=np.int64)
    n = lib().oracle_astar_2d(dimx, dimy, m.ctypes.data_as(U8P), start[0], start[1], goal[0], goal[1],
                              states.ctypes.data_as(I32P), cap, expanded.ctypes.data_as(I64P))
    return states[:n].tolist(), int(expanded[0])


def prioritized_sipp(inst, cap=1024):
    obst, obst_p = _i32(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
    starts, starts_p = _i32(inst["starts"])
    goals, goals_p = _i32(inst["goals"])
    n = len(starts)
    stats = np.zeros(4, dtype=np.int64)
    planned = np.zeros(n, dtype=np.int32)
    nst = np.zeros(n, dtype=np.int32)
    states = np.zeros((n, cap, 3), dtype=np.int32)
    k = lib().oracle_prioritized_sipp(inst["dimx"], inst["dimy"], len(obst), obst_p, n, starts_p, goals_p,
                                      stats.ctypes.data_as(I64P), planned.ctypes.data_as(I32P),
                                      nst.ctypes.data_as(I32P), states.ctypes.data_as(I32P), cap)
    return dict(n_planned=k, cost=int(stats[0]), expanded=int(stats[1]), planned=planned.tolist(),
                schedules=[states[i, :nst[i]].tolist() for i in range(n)])


def sipp_single(dimx, dimy, obstacles, start, goal, collision_intervals, cap=1024):
    """collision_intervals: [[x, y, start, end], ...] in file order."""
    obst, obst_p = _i32(np.asarray(obstacles, dtype=np.int32).reshape(-1, 2))
    ci, ci_p = _i32(np.asarray(collision_intervals, dtype=np.int32).reshape(-1, 4))
    states = np.zeros((cap, 3), dtype=np.int32)
    expanded = np.zeros(1, dtype=np.int64)
    n = lib().oracle_sipp_single(dimx, dimy, len(obst), obst_p, start[0], start[1], goal[0], goal[1], len(ci), ci_p,
                                 states.ctypes.data_as(I32P), cap, expanded.ctypes.data_as(I64P))
    return states[:n].tolist(), int(expanded[0])


def sipp_single_at(dimx, dimy, obstacles, start, goal, collision_intervals, start_time=0, cap=1024):
    """SIPP::search(start, wait, solution, startTime) (sipp.hpp:92): (states [x,y,t], expanded, cost, fmin)."""
    obst, obst_p = _i32(np.asarray(obstacles, dtype=np.int32).reshape(-1, 2))
    ci, ci_p = _i32(np.asarray(collision_intervals, dtype=np.int32).reshape(-1, 4))
    states = np.zeros((cap, 3), dtype=np.int32)
    expanded = np.zeros(1, dtype=np.int64)
    cf = np.zeros(2, dtype=np.int32)
    n = lib().oracle_sipp_single_at(dimx, dimy, len(obst), obst_p, start[0], start[1], goal[0], goal[1], len(ci), ci_p,
                                    start_time, states.ctypes.data_as(I32P), cap, expanded.ctypes.data_as(I64P),
                                    cf.ctypes.data_as(I32P))
    return states[:n].tolist(), int(expanded[0]), int(cf[0]), int(cf[1])


def mapf_solve_batch(algo, dimx, dimy, obstacles, starts, goals, w=1.0, cap_total=-1, n_threads=1):
    """n instances of one shape (int32 arrays [n][n_obst][2], [n][n_agents][2] x 2), one per thread on n_threads.
    Returns (per-instance int64 array [n][6] = rc, cost, makespan, hl, ll, elapsed_ns ; pool wall seconds)."""
    ob = np.ascontiguousarray(obstacles, dtype=np.int32)
    st = np.ascontiguousarray(starts, dtype=np.int32)
    go = np.ascontiguousarray(goals, dtype=np.int32)
    n = len(st)
    out = np.zeros((n, 6), dtype=np.int64)
    wall = lib().oracle_mapf_solve_batch(algo, w, n, dimx, dimy, ob.shape[1], ob.ctypes.data_as(I32P), st.shape[1],
                                         st.ctypes.data_as(I32P), go.ctypes.data_as(I32P), cap_total, n_threads,
                                         out.ctypes.data_as(I64P))
    return out, wall / 1e9


def mapf_solve_batch_digest(algo, dimx, dimy, obstacles, starts, goals, w=1.0, cap_total=-1, n_threads=1):
    """mapf_solve_batch with a seventh word per instance: the FNV-1a (64 bit) digest of the schedule as
    include/mrp_hl.h mrp_hl_solution::schedule_digest defines it (as uint64 in the returned second array)."""
    ob = np.ascontiguousarray(obstacles, dtype=np.int32)
    st = np.ascontiguousarray(starts, dtype=np.int32)
    go = np.ascontiguousarray(goals, dtype=np.int32)
    n = len(st)
    out = np.zeros((n, 7), dtype=np.int64)
    wall = lib().oracle_mapf_solve_batch_digest(algo, w, n, dimx, dimy, ob.shape[1], ob.ctypes.data_as(I32P), st.shape[1],
                                                st.ctypes.data_as(I32P), go.ctypes.data_as(I32P), cap_total, n_threads,
                                                out.ctypes.data_as(I64P))
    return out[:, :6], out[:, 6].view(np.uint64), wall / 1e9


def conflict_scan(paths):
    """getFirstConflict + focalHeuristic (ecbs.cpp:401-452, :315-350) of one solution: paths = [[[x, y], ...], ...]."""
    lens, lens_p = _i32([len(p) for p in paths])
    xy, xy_p = _i32(np.asarray([c for p in paths for c in p], dtype=np.int32).reshape(-1, 2))
    out = np.zeros(10, dtype=np.int32)
    lib().oracle_conflict_scan(len(paths), lens_p, xy_p, out.ctypes.data_as(I32P))
    keys = ("found", "time", "agent1", "agent2", "type", "x1", "y1", "x2", "y2", "count")
    return dict(zip(keys, (int(v) for v in out)))


def prioritized_sipp_batch(dimx, dimy, obstacles, starts, goals, n_threads=1):
    """n instances of one shape; returns (int64 array [n][4] = n_planned, cost, expanded, elapsed_ns ; pool wall seconds)."""
    ob = np.ascontiguousarray(obstacles, dtype=np.int32)
    st = np.ascontiguousarray(starts, dtype=np.int32)
    go = np.ascontiguousarray(goals, dtype=np.int32)
    n = len(st)
    out = np.zeros((n, 4), dtype=np.int64)
    wall = lib().oracle_prioritized_sipp_batch(n, dimx, dimy, ob.shape[1], ob.ctypes.data_as(I32P), st.shape[1],
                                               st.ctypes.data_as(I32P), go.ctypes.data_as(I32P), n_threads,
                                               out.ctypes.data_as(I64P))
    return out, wall / 1e9


def ta_ll_search(inst_map, start, goal, vertex_constraints=(), edge_constraints=(), cap_expansions=-1, cap=1024):
    """One low-level search of the task-assignment callers (example/cbs_ta.cpp's Environment under AStar): goal = None for an
    agent without a task.  Returns success, cost, fmin, expanded, states [t, x, y], actions, action_costs."""
    obst, obst_p = _i32(np.asarray(inst_map["obstacles"], dtype=np.int32).reshape(-1, 2))
    vc, vc_p = _i32(np.asarray(vertex_constraints, dtype=np.int32).reshape(-1, 3))
    ec, ec_p = _i32(np.asarray(edge_constraints, dtype=np.int32).reshape(-1, 5))
    out = np.zeros(4, dtype=np.int32)
    expanded = np.zeros(1, dtype=np.int64)
    states = np.zeros((cap, 3), dtype=np.int32)
    actions = np.zeros(cap, dtype=np.int32)
    costs = np.zeros(cap, dtype=np.int32)
    g = goal if goal is not None else (0, 0)
    rc = lib().oracle_ta_ll_search(inst_map["dimx"], inst_map["dimy"], len(obst), obst_p, start[0], start[1],
                                   0 if goal is None else 1, g[0], g[1], len(vc), vc_p, len(ec), ec_p, cap_expansions,
                                   out.ctypes.data_as(I32P), expanded.ctypes.data_as(I64P), states.ctypes.data_as(I32P),
                                   actions.ctypes.data_as(I32P), costs.ctypes.data_as(I32P), cap)
    n = int(out[3])
    assert n <= cap
    return dict(rc=rc, success=bool(out[0]), cost=int(out[1]), fmin=int(out[2]), expanded=int(expanded[0]),
                states=states[:n].tolist(), actions=actions[:max(n - 1, 0)].tolist(),
                action_costs=costs[:max(n - 1, 0)].tolist())


def ta_cbs_fixed(inst_map, starts, tasks):
    """cbs_ta.hpp's conflict tree for ONE fixed assignment (tasks[i] = [x, y] or None).  Returns (summary, low-level calls)."""
    obst, obst_p = _i32(np.asarray(inst_map["obstacles"], dtype=np.int32).reshape(-1, 2))
    st, st_p = _i32(starts)
    tk, tk_p = _i32([t if t is not None else [-1, -1] for t in tasks])
    n = len(starts)
    stats = np.zeros(3, dtype=np.int64)
    end = np.zeros((n, 3), dtype=np.int32)
    ncalls = np.zeros(1, dtype=np.int32)
    words = 1 << 16
    while True:
        buf = np.zeros(words, dtype=np.int32)
        need = lib().oracle_ta_cbs_fixed(inst_map["dimx"], inst_map["dimy"], len(obst), obst_p, n, st_p, tk_p,
                                         stats.ctypes.data_as(I64P), end.ctypes.data_as(I32P), buf.ctypes.data_as(I32P), words,
                                         ncalls.ctypes.data_as(I32P))
        if need <= words:
            break
        words = int(need)
    calls = []
    p = 0
    for _ in range(int(ncalls[0])):
        agent, has, tx, ty, ok, cost, fmin, expanded, nvc, nec, nst = (int(v) for v in buf[p:p + 11])
        p += 11
        vc = buf[p:p + 3 * nvc].reshape(-1, 3).tolist(); p += 3 * nvc
        ec = buf[p:p + 5 * nec].reshape(-1, 5).tolist(); p += 5 * nec
        states = buf[p:p + 3 * nst].reshape(-1, 3).tolist(); p += 3 * nst
        costs = buf[p:p + max(nst - 1, 0)].tolist(); p += max(nst - 1, 0)
        calls.append(dict(agent=agent, goal=[tx, ty] if has else None, success=bool(ok), cost=cost, fmin=fmin, expanded=expanded,
                          vertex_constraints=vc, edge_constraints=ec, states=states, action_costs=costs))
    return dict(solved=bool(stats[0]), cost=int(stats[1]), end=end.tolist()), calls
