"""ctypes binding of include/mrp_ll.h — the C-ABI of the HIP low-level search engine.

This is plumbing for tests and benchmarks; it mirrors the reference's low-level call
``LowLevelSearch_t(llenv[, w]).search(start, out)`` (cbs.hpp:99-101,155-157; ecbs.hpp:126-129,265-268) as
``LowLevelEngine.search_batch([LLJob...]) -> [LLResult...]``.  There is no CPU fallback: if the HIP library is
missing or no GPU is visible, construction raises.
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MRP_LL_LIB") or os.path.join(_PKG, "lib", "libmrp_ll.so")

ASTAR, ASTAR_EPS, SIPP, ASTAR_TA = 0, 1, 2, 3
JOB_STORE_RESULT, JOB_NO_GOAL, JOB_ROOT_CHAIN, JOB_HEAVY = 1, 2, 4, 8  # mrp_ll_job.flags (include/mrp_ll.h)
OK, NO_SOLUTION, CAP_EXPANSIONS, CAP_NODES, CAP_HORIZON, BAD_JOB, PATH_TRUNCATED, CAP_FOCAL = range(8)
ACTION_NAMES = ["Up", "Down", "Left", "Right", "Wait"]  # example/ecbs.cpp:49-55

I32P = ctypes.POINTER(ctypes.c_int32)


class mrp_ll_options(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("n_tickets", ctypes.c_int32), ("slots", ctypes.c_int32),
                ("arena_nodes", ctypes.c_int32), ("max_horizon", ctypes.c_int32), ("max_cells", ctypes.c_int32),
                ("lds_nodes", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class mrp_ll_job(ctypes.Structure):
    _fields_ = [("map_id", ctypes.c_int32), ("algo", ctypes.c_int32), ("w", ctypes.c_float),
                ("agent_idx", ctypes.c_int32), ("start_x", ctypes.c_int32), ("start_y", ctypes.c_int32),
                ("goal_x", ctypes.c_int32), ("goal_y", ctypes.c_int32),
                ("n_vertex_constraints", ctypes.c_int32), ("vertex_constraints", I32P),
                ("n_edge_constraints", ctypes.c_int32), ("edge_constraints", I32P),
                ("n_agents", ctypes.c_int32), ("path_len", I32P), ("path_xy", ctypes.POINTER(I32P)),
                ("max_expansions", ctypes.c_int64),
                ("n_collision_locations", ctypes.c_int32), ("collision_xy", I32P), ("collision_count", I32P),
                ("collision_intervals", I32P), ("initial_cost", ctypes.c_int32), ("sipp_commit", ctypes.c_int32),
                ("sipp_table", ctypes.c_void_p), ("path_ids", I32P), ("result_path_id", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("heuristic_id", ctypes.c_int32), ("chain_count", ctypes.c_int32),
                ("chain_starts_goals_xy", I32P)]


class mrp_ll_result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("cost", ctypes.c_int32), ("fmin", ctypes.c_int32),
                ("n_states", ctypes.c_int32), ("expanded", ctypes.c_int64), ("states_txy", I32P),
                ("actions", I32P), ("states_cap", ctypes.c_int32), ("tier", ctypes.c_int32),
                ("action_costs", I32P), ("chain_results", ctypes.c_void_p)]


class mrp_ll_conflict(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("found", "time", "agent1", "agent2", "type", "x1", "y1", "x2", "y2", "count")]


class mrp_ll_stats(ctypes.Structure):
    _fields_ = [("launches", ctypes.c_int64), ("jobs", ctypes.c_int64), ("expansions", ctypes.c_int64),
                ("nodes_created", ctypes.c_int64), ("migrated", ctypes.c_int64), ("kernel_ms", ctypes.c_double),
                ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double), ("session_busy_ms", ctypes.c_double),
                ("session_idle_ms", ctypes.c_double), ("session_active_wgs", ctypes.c_int64), ("pack_ms", ctypes.c_double),
                ("unpack_ms", ctypes.c_double), ("staged_bytes", ctypes.c_int64), ("prof", ctypes.c_int64 * 8),
                ("heavy_busy_ms", ctypes.c_double), ("heavy_idle_ms", ctypes.c_double), ("heavy_active_wgs", ctypes.c_int64),
                ("heavy_fallbacks", ctypes.c_int64)]


EXPORTS = ["mrp_ll_create", "mrp_ll_destroy", "mrp_ll_last_error", "mrp_ll_upload_map", "mrp_ll_search_batch",
           "mrp_ll_submit", "mrp_ll_wait", "mrp_ll_get_stats", "mrp_ll_reset_stats", "mrp_ll_version",
           "mrp_ll_session_begin", "mrp_ll_session_end", "mrp_ll_poll", "mrp_ll_poll_any", "mrp_ll_submit_lane", "mrp_ll_sync_maps",
           "mrp_ll_configure_tiers", "mrp_ll_session_occupancy", "mrp_ll_session_begin_sipp", "mrp_ll_release_maps", "mrp_ll_session_begin_algo", "mrp_ll_conflict_scan",
           "mrp_ll_sipp_table_create", "mrp_ll_sipp_table_add", "mrp_ll_sipp_table_destroy", "mrp_ll_path_store_reserve",
           "mrp_ll_upload_heuristic", "mrp_ll_session_begin_tiers", "mrp_ll_session_tiers_geometry",
           "mrp_ll_session_begin_tiers_gated", "mrp_ll_submit_tagged", "mrp_ll_poll_any_tagged"]

_lib = None


def load_library(path: Optional[str] = None):
    """Load libmrp_ll.so (raises OSError with a build hint if it is missing — no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `python -m libmultirobotplanning_amd._build` "
                      "(hipcc --offload-arch=gfx950); the engine has no CPU fallback")
    lib = ctypes.CDLL(p)
    lib.mrp_ll_create.restype = ctypes.c_int
    lib.mrp_ll_create.argtypes = [ctypes.POINTER(mrp_ll_options), ctypes.POINTER(ctypes.c_void_p)]
    lib.mrp_ll_destroy.restype = None
    lib.mrp_ll_destroy.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_last_error.restype = ctypes.c_char_p
    lib.mrp_ll_last_error.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_upload_map.restype = ctypes.c_int
    lib.mrp_ll_upload_map.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, I32P, I32P]
    lib.mrp_ll_search_batch.restype = ctypes.c_int
    lib.mrp_ll_search_batch.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(mrp_ll_job),
                                        ctypes.POINTER(mrp_ll_result)]
    lib.mrp_ll_submit.restype = ctypes.c_int
    lib.mrp_ll_submit.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(mrp_ll_job),
                                  ctypes.POINTER(mrp_ll_result), I32P]
    lib.mrp_ll_wait.restype = ctypes.c_int
    lib.mrp_ll_wait.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mrp_ll_get_stats.restype = ctypes.c_int
    lib.mrp_ll_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(mrp_ll_stats)]
    lib.mrp_ll_reset_stats.restype = ctypes.c_int
    lib.mrp_ll_reset_stats.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_version.restype = ctypes.c_char_p
    lib.mrp_ll_version.argtypes = []
    lib.mrp_ll_session_begin.restype = ctypes.c_int
    lib.mrp_ll_session_begin.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mrp_ll_session_begin_algo.restype = ctypes.c_int
    lib.mrp_ll_session_begin_algo.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.mrp_ll_session_begin_tiers.restype = ctypes.c_int
    lib.mrp_ll_session_begin_tiers.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.mrp_ll_session_tiers_geometry.restype = ctypes.c_int
    lib.mrp_ll_session_tiers_geometry.argtypes = [ctypes.c_void_p, I32P, I32P, I32P]
    lib.mrp_ll_session_end.restype = ctypes.c_int
    lib.mrp_ll_session_end.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_poll.restype = ctypes.c_int
    lib.mrp_ll_poll.argtypes = [ctypes.c_void_p, ctypes.c_int32, I32P]
    lib.mrp_ll_submit_lane.restype = ctypes.c_int
    lib.mrp_ll_submit_lane.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(mrp_ll_job),
                                       ctypes.POINTER(mrp_ll_result), I32P]
    lib.mrp_ll_sync_maps.restype = ctypes.c_int
    lib.mrp_ll_sync_maps.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_release_maps.restype = ctypes.c_int
    lib.mrp_ll_release_maps.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_conflict_scan.restype = ctypes.c_int
    lib.mrp_ll_conflict_scan.argtypes = [ctypes.c_void_p, ctypes.c_int32, I32P, I32P, I32P, ctypes.POINTER(mrp_ll_conflict)]
    lib.mrp_ll_path_store_reserve.restype = ctypes.c_int
    lib.mrp_ll_upload_heuristic.restype = ctypes.c_int
    lib.mrp_ll_upload_heuristic.argtypes = [ctypes.c_void_p, ctypes.c_int32, I32P, ctypes.POINTER(ctypes.c_int32)]
    lib.mrp_ll_path_store_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mrp_ll_sipp_table_create.restype = ctypes.c_int
    lib.mrp_ll_sipp_table_create.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    lib.mrp_ll_sipp_table_add.restype = ctypes.c_int
    lib.mrp_ll_sipp_table_add.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 4
    lib.mrp_ll_sipp_table_destroy.restype = None
    lib.mrp_ll_sipp_table_destroy.argtypes = [ctypes.c_void_p]
    lib.mrp_ll_poll_any.restype = ctypes.c_int
    lib.mrp_ll_poll_any.argtypes = [ctypes.c_void_p, I32P, ctypes.c_int32, I32P]
    if path is None:
        _lib = lib
    return lib


@dataclass
class LLJob:
    """One low-level search: LowLevelEnvironment(env, agent, constraints[, solution]) + search(start, out)."""
    map_id: int
    algo: int
    start: Sequence[int]
    goal: Optional[Sequence[int]]  # None: ASTAR_TA for an agent without a task (MRP_LL_JOB_NO_GOAL)
    agent_idx: int = 0
    w: float = 1.0
    vertex_constraints: Sequence[Sequence[int]] = ()   # (time, x, y)
    edge_constraints: Sequence[Sequence[int]] = ()     # (time, x1, y1, x2, y2)
    ctx_paths: Sequence[Sequence[Sequence[int]]] = ()  # per agent [[x, y], ...]; [] = empty path
    max_expansions: int = -1
    collision_intervals: Sequence[Sequence[int]] = ()  # SIPP: [x, y, start, end] (grouped per location, in order)
    initial_cost: int = 0  # A*: AStar::search(..., initialCost) a_star.hpp:64; SIPP: SIPP::search(..., startTime) sipp.hpp:92
    sipp_table: Optional[int] = None  # SIPP: handle from LowLevelEngine.sipp_table_create (replaces collision_intervals)
    sipp_commit: bool = False         # SIPP with sipp_table: on success the path's stays join the table (mrp_ll.h)
    path_ids: Optional[Sequence[int]] = None  # f2: path-store slots of ctx_paths (-1 = none); lengths come from ctx_paths
    result_path_id: int = -1                  # f2: path-store slot that also receives the result path
    heuristic_id: int = -1                    # ASTAR_TA: LowLevelEngine.upload_heuristic of the goal cell
    heavy: bool = False                       # MRP_LL_JOB_HEAVY: the search is known to outgrow the LDS tier (a hint)


@dataclass
class LLResult:
    status: int
    success: bool
    cost: int
    fmin: int
    expanded: int
    states: List[List[int]] = field(default_factory=list)   # [t, x, y]
    actions: List[int] = field(default_factory=list)
    tier: int = 0
    action_costs: List[int] = field(default_factory=list)


class LowLevelEngine:
    def __init__(self, device: int = 0, n_tickets: int = 0, slots: int = 0, arena_nodes: int = 0,
                 max_horizon: int = 0, max_cells: int = 0, lds_nodes: int = 0):
        self._lib = load_library()
        opt = mrp_ll_options(device, n_tickets, slots, arena_nodes, max_horizon, max_cells, lds_nodes, 0)
        h = ctypes.c_void_p()
        rc = self._lib.mrp_ll_create(ctypes.byref(opt), ctypes.byref(h))
        if rc != 0 or not h:
            raise RuntimeError(f"mrp_ll_create failed (rc={rc}): a HIP device is required, there is no CPU fallback")
        self._h = h
        self.max_horizon = max_horizon or 512

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mrp_ll_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}: {self._lib.mrp_ll_last_error(self._h).decode()}")

    def upload_map(self, dimx, dimy, obstacles) -> int:
        ob = np.ascontiguousarray(np.asarray(obstacles, dtype=np.int32).reshape(-1, 2))
        mid = ctypes.c_int32(-1)
        self._check(self._lib.mrp_ll_upload_map(self._h, dimx, dimy, len(ob), ob.ctypes.data_as(I32P),
                                                ctypes.byref(mid)), "mrp_ll_upload_map")
        return mid.value

    def _marshal(self, jobs: Sequence[LLJob], cap: int):
        """Build the ctypes job/result arrays for `jobs`; returns (cjobs, cres, keepalive)."""
        n = len(jobs)
        cjobs = (mrp_ll_job * max(n, 1))()
        cres = (mrp_ll_result * max(n, 1))()
        keep = []
        states = np.zeros((max(n, 1), cap, 3), dtype=np.int32)
        actions = np.zeros((max(n, 1), cap), dtype=np.int32)
        costs = np.zeros((max(n, 1), cap), dtype=np.int32)
        for i, j in enumerate(jobs):
            cj = cjobs[i]
            cj.map_id, cj.algo, cj.w, cj.agent_idx = j.map_id, j.algo, j.w, j.agent_idx
            cj.start_x, cj.start_y = j.start[0], j.start[1]
            cj.goal_x, cj.goal_y = (j.goal[0], j.goal[1]) if j.goal is not None else (0, 0)
            vc = np.ascontiguousarray(np.asarray(j.vertex_constraints, dtype=np.int32).reshape(-1, 3))
            ec = np.ascontiguousarray(np.asarray(j.edge_constraints, dtype=np.int32).reshape(-1, 5))
            cj.n_vertex_constraints, cj.vertex_constraints = len(vc), vc.ctypes.data_as(I32P)
            cj.n_edge_constraints, cj.edge_constraints = len(ec), ec.ctypes.data_as(I32P)
            na = len(j.ctx_paths)
            plen = np.asarray([len(p) for p in j.ctx_paths], dtype=np.int32)
            parr = [np.ascontiguousarray(np.asarray(p, dtype=np.int32).reshape(-1, 2)) for p in j.ctx_paths]
            pptr = (I32P * max(na, 1))(*[a.ctypes.data_as(I32P) for a in parr])
            cj.n_agents = na
            cj.path_len = plen.ctypes.data_as(I32P)
            cj.path_xy = ctypes.cast(pptr, ctypes.POINTER(I32P))
            cj.max_expansions = j.max_expansions
            cj.initial_cost = j.initial_cost
            if j.sipp_table is not None:
                cj.sipp_table = j.sipp_table
                cj.sipp_commit = 1 if j.sipp_commit else 0
            cj.result_path_id = j.result_path_id
            cj.flags = (JOB_STORE_RESULT if j.result_path_id >= 0 else 0) | (JOB_NO_GOAL if j.goal is None else 0) | \
                (JOB_HEAVY if j.heavy else 0)
            cj.heuristic_id = j.heuristic_id
            if j.path_ids is not None:
                ids = np.ascontiguousarray(np.asarray(j.path_ids, dtype=np.int32))
                cj.path_ids = ids.ctypes.data_as(I32P)
                keep.append(ids)
            if j.collision_intervals:
                locs, counts, ivs = [], [], []
                for x, y, a, b in j.collision_intervals:  # consecutive entries of one location form one list
                    if locs and locs[-1] == [x, y]:
                        counts[-1] += 1
                    else:
                        locs.append([x, y])
                        counts.append(1)
                    ivs.append([a, b])
                cxy = np.ascontiguousarray(np.asarray(locs, dtype=np.int32))
                ccnt = np.ascontiguousarray(np.asarray(counts, dtype=np.int32))
                civ = np.ascontiguousarray(np.asarray(ivs, dtype=np.int32))
                cj.n_collision_locations = len(locs)
                cj.collision_xy, cj.collision_count = cxy.ctypes.data_as(I32P), ccnt.ctypes.data_as(I32P)
                cj.collision_intervals = civ.ctypes.data_as(I32P)
                keep.append((cxy, ccnt, civ))
            keep.append((vc, ec, plen, parr, pptr))
            cres[i].states_txy = states[i].ctypes.data_as(I32P)
            cres[i].actions = actions[i].ctypes.data_as(I32P)
            cres[i].states_cap = cap
            cres[i].action_costs = costs[i].ctypes.data_as(I32P)
        return cjobs, cres, (keep, states, actions, costs)

    def search_batch(self, jobs: Sequence[LLJob], states_cap: Optional[int] = None) -> List[LLResult]:
        n = len(jobs)
        cap = states_cap or self.max_horizon
        cjobs, cres, (keep, states, actions, costs) = self._marshal(jobs, cap)
        self._check(self._lib.mrp_ll_search_batch(self._h, n, cjobs, cres), "mrp_ll_search_batch")
        out = []
        for i in range(n):
            r = cres[i]
            ns = r.n_states if r.status in (OK, PATH_TRUNCATED) else 0
            m = min(ns, cap)
            out.append(LLResult(status=r.status, success=r.status in (OK, PATH_TRUNCATED), cost=r.cost, fmin=r.fmin,
                                expanded=r.expanded, states=states[i, :m].tolist(),
                                actions=actions[i, :max(m - 1, 0)].tolist(), tier=r.tier,
                                action_costs=costs[i, :max(m - 1, 0)].tolist()))
        return out

    def conflict_scan(self, solutions: Sequence[Sequence[Sequence[Sequence[int]]]]) -> List[dict]:
        """getFirstConflict (example/ecbs.cpp:401-452) + focalHeuristic (:315-350) for a batch of solutions, each a list
        of paths [[x, y], ...].  Returns per solution dict(found, time, agent1, agent2, type, x1, y1, x2, y2, count)."""
        n = len(solutions)
        set_first = np.zeros(n + 1, dtype=np.int32)
        lens = []
        for s, sol in enumerate(solutions):
            set_first[s + 1] = set_first[s] + len(sol)
            lens.extend(len(p) for p in sol)
        path_first = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(np.asarray(lens, dtype=np.int64), out=path_first[1:])
        flat = [xy for sol in solutions for p in sol for xy in p]
        xy = np.ascontiguousarray(np.asarray(flat, dtype=np.int32).reshape(-1, 2))
        out = (mrp_ll_conflict * max(n, 1))()
        self._check(self._lib.mrp_ll_conflict_scan(self._h, n, set_first.ctypes.data_as(I32P),
                                                   path_first.ctypes.data_as(I32P), xy.ctypes.data_as(I32P), out),
                    "mrp_ll_conflict_scan")
        return [{k: getattr(out[i], k) for k, _ in mrp_ll_conflict._fields_} for i in range(n)]

    def upload_heuristic(self, map_id: int, dist) -> int:
        """MRP_LL_ASTAR_TA: the shortest-path table of one goal cell, dist[dimy][dimx] (INT32_MAX = unreachable)."""
        d = np.ascontiguousarray(np.asarray(dist, dtype=np.int64).clip(-1, 2 ** 31 - 1).astype(np.int32).reshape(-1))
        hid = ctypes.c_int32(-1)
        self._check(self._lib.mrp_ll_upload_heuristic(self._h, map_id, d.ctypes.data_as(I32P), ctypes.byref(hid)),
                    "mrp_ll_upload_heuristic")
        return hid.value

    def path_store_reserve(self, n_slots: int) -> None:
        """Allocate the device-resident path store (f2): slots 0..n_slots-1 are the caller's to hand out."""
        self._check(self._lib.mrp_ll_path_store_reserve(self._h, n_slots), "mrp_ll_path_store_reserve")

    def sipp_table_create(self, map_id: int) -> int:
        h = ctypes.c_void_p()
        self._check(self._lib.mrp_ll_sipp_table_create(self._h, map_id, ctypes.byref(h)), "mrp_ll_sipp_table_create")
        return h.value

    def sipp_table_add(self, table: int, x: int, y: int, start: int, end: int) -> None:
        self._check(self._lib.mrp_ll_sipp_table_add(table, x, y, start, end), "mrp_ll_sipp_table_add")

    def sipp_table_destroy(self, table: int) -> None:
        self._lib.mrp_ll_sipp_table_destroy(table)

    def session_occupancy(self, algo: int) -> int:
        """mrp_ll_session_occupancy: resident searches per CU of a session of `algo`."""
        occ = ctypes.c_int32(0)
        self._check(self._lib.mrp_ll_session_occupancy(self._h, algo, ctypes.byref(occ)), "mrp_ll_session_occupancy")
        return occ.value

    def configure_tiers(self, lds_nodes: int = 0, lds_rows: int = 0, lds_path_bytes: int = 0) -> int:
        """Geometry of the LDS fast tier for the launches that follow (0 = keep); returns resident searches per CU."""
        occ = ctypes.c_int32(0)
        self._check(self._lib.mrp_ll_configure_tiers(self._h, lds_nodes, lds_rows, lds_path_bytes, ctypes.byref(occ)),
                    "mrp_ll_configure_tiers")
        return occ.value

    def session_begin(self, workgroups: int = 0):
        """Keep `workgroups` wavefronts resident and feed them through the host job ring (see mrp_ll.h)."""
        self._check(self._lib.mrp_ll_session_begin(self._h, workgroups), "mrp_ll_session_begin")

    def session_begin_algo(self, algo: int, workgroups: int = 0):
        """A session for jobs of one algorithm only (the specialised resident kernel)."""
        self._check(self._lib.mrp_ll_session_begin_algo(self._h, algo, workgroups), "mrp_ll_session_begin_algo")

    def session_begin_tiers(self, workgroups: int, heavy_workgroups: int):
        """An A*-epsilon session as a pair of resident launches: front workgroups (LDS tier only) + heavy workgroups (wide
        LDS tier, arena tier) that take over the searches that outgrow the front tier (include/mrp_ll.h)."""
        self._check(self._lib.mrp_ll_session_begin_tiers(self._h, ASTAR_EPS, workgroups, heavy_workgroups),
                    "mrp_ll_session_begin_tiers")

    def session_begin_sipp(self, workgroups: int = 0):
        """A session for MRP_LL_SIPP jobs (resident SIPP kernel); other jobs come back as BAD_JOB."""
        self._check(self._lib.mrp_ll_session_begin_sipp(self._h, workgroups), "mrp_ll_session_begin_sipp")

    def session_end(self):
        self._check(self._lib.mrp_ll_session_end(self._h), "mrp_ll_session_end")

    def stats(self) -> dict:
        st = mrp_ll_stats()
        self._check(self._lib.mrp_ll_get_stats(self._h, ctypes.byref(st)), "mrp_ll_get_stats")
        return {k: (list(getattr(st, k)) if k == "prof" else getattr(st, k)) for k, _ in mrp_ll_stats._fields_}

    def reset_stats(self):
        self._check(self._lib.mrp_ll_reset_stats(self._h), "mrp_ll_reset_stats")
