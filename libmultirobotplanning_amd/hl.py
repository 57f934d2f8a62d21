"""ctypes binding of include/mrp_hl.h — the host-side CBS / ECBS conflict-tree drivers that call the HIP engine.

Mirrors the reference's top-level calls ``CBS(env).search(starts, solution)`` (cbs.hpp:85) and
``ECBS(env, w).search(starts, solution)`` (ecbs.hpp:109) for a whole batch of instances at once, and returns the
``statistics:`` fields of example/ecbs.cpp:594-599 per instance.  No CPU fallback.
"""
import ctypes
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import ll as _ll

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "lib", "libmrp_hl.so")

CBS, ECBS = 0, 1
SOLVED, NO_SOLUTION, CAP, LL_ERROR = 0, 1, 2, 3
I32P = ctypes.POINTER(ctypes.c_int32)


class mrp_hl_instance(ctypes.Structure):
    _fields_ = [("dimx", ctypes.c_int32), ("dimy", ctypes.c_int32), ("n_obstacles", ctypes.c_int32),
                ("obstacles_xy", I32P), ("n_agents", ctypes.c_int32), ("starts_xy", I32P), ("goals_xy", I32P)]


class mrp_hl_solution(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("n_ll_searches", ctypes.c_int32), ("cost", ctypes.c_int64),
                ("makespan", ctypes.c_int64), ("high_level_expanded", ctypes.c_int64),
                ("low_level_expanded", ctypes.c_int64), ("path_len", I32P), ("paths_xy", I32P),
                ("path_cap", ctypes.c_int32), ("reserved", ctypes.c_int32), ("schedule_digest", ctypes.c_uint64)]


# the same record as a numpy dtype (bulk access to arrays of results)
_SOLUTION_DTYPE = np.dtype([("status", "<i4"), ("n_ll_searches", "<i4"), ("cost", "<i8"), ("makespan", "<i8"),
                            ("high_level_expanded", "<i8"), ("low_level_expanded", "<i8"), ("path_len", "<u8"),
                            ("paths_xy", "<u8"), ("path_cap", "<i4"), ("reserved", "<i4"), ("schedule_digest", "<u8")])


class mrp_hl_options(ctypes.Structure):
    _fields_ = [("algo", ctypes.c_int32), ("w", ctypes.c_float), ("max_ll_expansions", ctypes.c_int64),
                ("max_hl_expansions", ctypes.c_int64), ("n_threads", ctypes.c_int32), ("mode", ctypes.c_int32)]


class mrp_hl_batch_stats(ctypes.Structure):
    _fields_ = [("wall_seconds", ctypes.c_double), ("rounds", ctypes.c_int64), ("ll_searches", ctypes.c_int64),
                ("ll_expansions", ctypes.c_int64), ("solved", ctypes.c_int64), ("build_seconds", ctypes.c_double),
                ("ll_call_seconds", ctypes.c_double), ("consume_seconds", ctypes.c_double),
                ("speculative_searches", ctypes.c_int64), ("wasted_ll_expansions", ctypes.c_int64),
                ("root_solved", ctypes.c_int64)]


class mrp_hl_sipp_solution(ctypes.Structure):
    _fields_ = [("cost", ctypes.c_int64), ("low_level_expanded", ctypes.c_int64), ("n_planned", ctypes.c_int32),
                ("status", ctypes.c_int32), ("planned", I32P), ("n_states", I32P), ("states_xyt", I32P),
                ("state_cap", ctypes.c_int32), ("reserved2", ctypes.c_int32)]


EXPORTS = ["mrp_hl_solver_prioritized_sipp", "mrp_hl_solver_preload", "mrp_hl_solver_solve_preloaded",
           "mrp_hl_solver_solve_stream",
           "mrp_hl_preloaded_free", "mrp_hl_solve_batch", "mrp_hl_solver_create", "mrp_hl_solver_destroy", "mrp_hl_solver_solve",
           "mrp_hl_solver_ll_stats", "mrp_hl_solver_last_error", "mrp_hl_generate_instance", "mrp_hl_generate_instances", "mrp_hl_astar_grid2d",
           "mrp_hl_ct_create", "mrp_hl_ct_destroy", "mrp_hl_ct_n_requests", "mrp_hl_ct_request", "mrp_hl_ct_deliver",
           "mrp_hl_ct_done", "mrp_hl_ct_solution"]

_lib = None


def load_library(path: Optional[str] = None):
    """Load libmrp_hl.so.  `path` is for tests that bind the same ABI from another build of the drivers."""
    global _lib
    if _lib is None or path is not None:
        if path is None:
            _ll.load_library()  # libmrp_hl.so links against libmrp_ll.so
            if not os.path.exists(_LIB_PATH):
                raise OSError(f"{_LIB_PATH} not found: build it with `python -m libmultirobotplanning_amd._build`")
        lib = ctypes.CDLL(path or _LIB_PATH)
        lib.mrp_hl_solver_create.restype = ctypes.c_int
        lib.mrp_hl_solver_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_ll.mrp_ll_options),
                                             ctypes.POINTER(ctypes.c_void_p)]
        lib.mrp_hl_solver_destroy.restype = None
        lib.mrp_hl_solver_destroy.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_solver_solve.restype = ctypes.c_int
        lib.mrp_hl_solver_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(mrp_hl_options), ctypes.c_int32,
                                            ctypes.POINTER(mrp_hl_instance), ctypes.POINTER(mrp_hl_solution),
                                            ctypes.POINTER(mrp_hl_batch_stats)]
        lib.mrp_hl_solver_ll_stats.restype = ctypes.c_int
        lib.mrp_hl_solver_ll_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ll.mrp_ll_stats), ctypes.c_int32]
        lib.mrp_hl_solver_last_error.restype = ctypes.c_char_p
        lib.mrp_hl_solver_last_error.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_solve_batch.restype = ctypes.c_int
        lib.mrp_hl_solve_batch.argtypes = [ctypes.c_int32, ctypes.POINTER(mrp_hl_options), ctypes.c_int32,
                                           ctypes.POINTER(mrp_hl_instance), ctypes.POINTER(mrp_hl_solution),
                                           ctypes.POINTER(mrp_hl_batch_stats)]
        lib.mrp_hl_solver_preload.restype = ctypes.c_int
        lib.mrp_hl_solver_preload.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.POINTER(mrp_hl_instance), ctypes.POINTER(ctypes.c_void_p)]
        lib.mrp_hl_solver_solve_stream.restype = ctypes.c_int
        lib.mrp_hl_solver_solve_stream.argtypes = [ctypes.c_void_p, ctypes.POINTER(mrp_hl_options), ctypes.c_int32,
                                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                   ctypes.POINTER(mrp_hl_batch_stats)]
        lib.mrp_hl_solver_solve_preloaded.restype = ctypes.c_int
        lib.mrp_hl_solver_solve_preloaded.argtypes = [ctypes.c_void_p, ctypes.POINTER(mrp_hl_options), ctypes.c_void_p,
                                                      ctypes.POINTER(mrp_hl_solution), ctypes.POINTER(mrp_hl_batch_stats)]
        lib.mrp_hl_preloaded_free.restype = None
        lib.mrp_hl_preloaded_free.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_solver_prioritized_sipp.restype = ctypes.c_int
        lib.mrp_hl_solver_prioritized_sipp.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(mrp_hl_instance),
                                                       ctypes.POINTER(mrp_hl_sipp_solution),
                                                       ctypes.POINTER(mrp_hl_batch_stats)]
        lib.mrp_hl_generate_instance.restype = ctypes.c_int
        lib.mrp_hl_generate_instance.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int32, I32P, I32P, I32P]
        lib.mrp_hl_astar_grid2d.restype = ctypes.c_int32
        lib.mrp_hl_astar_grid2d.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, I32P, ctypes.c_int32, I32P,
                                            ctypes.POINTER(ctypes.c_int64)]
        lib.mrp_hl_ct_create.restype = ctypes.c_int
        lib.mrp_hl_ct_create.argtypes = [ctypes.POINTER(mrp_hl_instance), ctypes.POINTER(mrp_hl_options), ctypes.c_int32,
                                         ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
        lib.mrp_hl_ct_destroy.restype = None
        lib.mrp_hl_ct_destroy.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_ct_n_requests.restype = ctypes.c_int32
        lib.mrp_hl_ct_n_requests.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_ct_request.restype = ctypes.c_int
        lib.mrp_hl_ct_request.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(_ll.mrp_ll_job), I32P, I32P]
        lib.mrp_hl_ct_deliver.restype = ctypes.c_int
        lib.mrp_hl_ct_deliver.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.POINTER(_ll.mrp_ll_result)]
        lib.mrp_hl_ct_done.restype = ctypes.c_int32
        lib.mrp_hl_ct_done.argtypes = [ctypes.c_void_p]
        lib.mrp_hl_ct_solution.restype = ctypes.c_int
        lib.mrp_hl_ct_solution.argtypes = [ctypes.c_void_p, ctypes.POINTER(mrp_hl_solution)]
        lib.mrp_hl_generate_instances.restype = ctypes.c_int
        lib.mrp_hl_generate_instances.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_int32, I32P, I32P, I32P]
        if path is not None:
            return lib
        _lib = lib
    return _lib


def generate_instance(seed: int, dimx: int = 32, dimy: int = 32, n_obstacles: int = 204, n_agents: int = 10) -> Dict:
    """Seeded synthetic 32x32_obst204-shaped instance (identical on every box)."""
    lib = load_library()
    ob = np.zeros((n_obstacles, 2), dtype=np.int32)
    st = np.zeros((n_agents, 2), dtype=np.int32)
    go = np.zeros((n_agents, 2), dtype=np.int32)
    rc = lib.mrp_hl_generate_instance(seed, dimx, dimy, n_obstacles, n_agents, ob.ctypes.data_as(I32P),
                                      st.ctypes.data_as(I32P), go.ctypes.data_as(I32P))
    if rc != 0:
        raise ValueError("instance generation failed")
    return dict(dimx=dimx, dimy=dimy, obstacles=ob.tolist(), starts=st.tolist(), goals=go.tolist())


def astar_grid2d(dimx: int, dimy: int, mask, start: Sequence[int], goal: Sequence[int], cap: int = 65536):
    """example/a_star.cpp on a [dimy][dimx] obstacle mask (host plumbing, BASELINE configs[0]): (states [[x, y]..] or [],
    cost, expanded)."""
    lib = load_library()
    m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).reshape(dimy, dimx))
    out = np.zeros((cap, 2), dtype=np.int32)
    cost = ctypes.c_int32(0)
    exp = ctypes.c_int64(0)
    n = lib.mrp_hl_astar_grid2d(dimx, dimy, m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), start[0], start[1], goal[0],
                                goal[1], out.ctypes.data_as(I32P), cap, ctypes.byref(cost), ctypes.byref(exp))
    if n < 0:
        raise ValueError("mrp_hl_astar_grid2d: bad argument")
    return out[:min(n, cap)].tolist(), cost.value, exp.value


class InstanceArrays:
    """n instances of one shape as three contiguous int32 arrays (what mrp_hl_generate_instances fills): obstacles
    [n][n_obstacles][2], starts / goals [n][n_agents][2].  Indexing gives the dict form used elsewhere."""

    def __init__(self, dimx, dimy, obstacles, starts, goals):
        self.dimx, self.dimy = dimx, dimy
        self.obstacles, self.starts, self.goals = obstacles, starts, goals

    def __len__(self):
        return len(self.starts)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return InstanceArrays(self.dimx, self.dimy, self.obstacles[k], self.starts[k], self.goals[k])
        return dict(dimx=self.dimx, dimy=self.dimy, obstacles=self.obstacles[k].tolist(), starts=self.starts[k].tolist(),
                    goals=self.goals[k].tolist())

    def __iter__(self):
        return (self[k] for k in range(len(self)))


def load_shipped_corpus(path: str) -> List[Dict]:
    """The 1000 shipped benchmark/32x32_obst204 inputs from tests/golden/shipped_32x32.npz (uint8 arrays per agent count),
    as (name, instance dict) pairs in the order agents10_ex0 .. agents100_ex99."""
    z = np.load(path)
    out = []
    for n in range(10, 101, 10):
        ob, st, go = z["obst%d" % n], z["starts%d" % n], z["goals%d" % n]
        for k in range(ob.shape[0]):
            out.append(("map_32by32_obst204_agents%d_ex%d" % (n, k),
                        dict(dimx=32, dimy=32, obstacles=ob[k].astype(int).tolist(), starts=st[k].astype(int).tolist(),
                             goals=go[k].astype(int).tolist())))
    return out


def generate_instances(seed0: int, n: int, dimx: int = 32, dimy: int = 32, n_obstacles: int = 204,
                       n_agents: int = 10) -> InstanceArrays:
    """Seeds seed0 .. seed0+n-1 in ONE native call (SURVEY.md §8d generator); identical to n generate_instance calls."""
    lib = load_library()
    ob = np.zeros((n, n_obstacles, 2), dtype=np.int32)
    st = np.zeros((n, n_agents, 2), dtype=np.int32)
    go = np.zeros((n, n_agents, 2), dtype=np.int32)
    rc = lib.mrp_hl_generate_instances(seed0, n, dimx, dimy, n_obstacles, n_agents, ob.ctypes.data_as(I32P),
                                       st.ctypes.data_as(I32P), go.ctypes.data_as(I32P))
    if rc != 0:
        raise ValueError("instance generation failed")
    return InstanceArrays(dimx, dimy, ob, st, go)


class BatchSolver:
    """Persistent solver: engines/arenas are created once; solve() may be called repeatedly."""

    def __init__(self, device: int = 0, n_threads: int = 0, slots: int = 0, arena_nodes: int = 0,
                 max_horizon: int = 0, lds_nodes: int = 0, max_cells: int = 0, _lib_path: Optional[str] = None):
        self._lib = load_library(_lib_path)
        opt = _ll.mrp_ll_options(device, 1, slots, arena_nodes, max_horizon, max_cells, lds_nodes, 0)
        h = ctypes.c_void_p()
        rc = self._lib.mrp_hl_solver_create(device, n_threads, ctypes.byref(opt), ctypes.byref(h))
        if rc != 0 or not h:
            raise RuntimeError(f"mrp_hl_solver_create failed (rc={rc}): a HIP device is required (no CPU fallback)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mrp_hl_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, instances: Sequence[Dict], n_threads: int = 0, want_paths: bool = True, path_cap: int = 512):
        """Marshal a batch and upload its static maps to HBM (outside any timed region: the reference constructs its
        Environment before it starts its Timer, example/ecbs.cpp:576-582).  Returns an opaque prepared batch."""
        n = len(instances)
        cin = (mrp_hl_instance * max(n, 1))()
        csol = (mrp_hl_solution * max(n, 1))()
        keep = []
        plen: List[Optional[np.ndarray]] = []
        pxy: List[Optional[np.ndarray]] = []
        if isinstance(instances, InstanceArrays):
            # bulk form: fill the descriptor array through one numpy view instead of n x 7 ctypes attribute stores
            ob, st, go = (np.ascontiguousarray(a) for a in (instances.obstacles, instances.starts, instances.goals))
            keep.append((ob, st, go))
            rec = np.dtype([("dimx", "<i4"), ("dimy", "<i4"), ("n_obstacles", "<i4"), ("_p0", "<i4"),
                            ("obstacles_xy", "<u8"), ("n_agents", "<i4"), ("_p1", "<i4"), ("starts_xy", "<u8"),
                            ("goals_xy", "<u8")])
            assert rec.itemsize == ctypes.sizeof(mrp_hl_instance)
            view = np.frombuffer(cin, dtype=rec, count=n)
            view["dimx"], view["dimy"] = instances.dimx, instances.dimy
            view["n_obstacles"], view["n_agents"] = ob.shape[1], st.shape[1]
            view["obstacles_xy"] = ob.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(ob.shape[1] * 8)
            view["starts_xy"] = st.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(st.shape[1] * 8)
            view["goals_xy"] = go.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(go.shape[1] * 8)
            if want_paths:  # one block for all schedules: [n][agents][path_cap][2] int32 + [n][agents] lengths
                na = st.shape[1]
                lens = np.zeros((n, na), dtype=np.int32)
                paths = np.empty((n, na, path_cap, 2), dtype=np.int32)
                keep.append((lens, paths))
                assert _SOLUTION_DTYPE.itemsize == ctypes.sizeof(mrp_hl_solution)
                sv = np.frombuffer(csol, dtype=_SOLUTION_DTYPE, count=n)
                sv["path_len"] = lens.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(na * 4)
                sv["paths_xy"] = paths.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(na * path_cap * 8)
                sv["path_cap"] = path_cap
                plen, pxy = lens, paths
            instances = ()
        for i, inst in enumerate(instances):
            ob = np.ascontiguousarray(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
            st = np.ascontiguousarray(np.asarray(inst["starts"], dtype=np.int32).reshape(-1, 2))
            go = np.ascontiguousarray(np.asarray(inst["goals"], dtype=np.int32).reshape(-1, 2))
            keep.append((ob, st, go))
            c = cin[i]
            c.dimx, c.dimy = inst["dimx"], inst["dimy"]
            c.n_obstacles, c.obstacles_xy = len(ob), ob.ctypes.data_as(I32P)
            c.n_agents, c.starts_xy, c.goals_xy = len(st), st.ctypes.data_as(I32P), go.ctypes.data_as(I32P)
            if want_paths:
                a = np.zeros(len(st), dtype=np.int32)
                b = np.zeros((len(st), path_cap, 2), dtype=np.int32)
                plen.append(a)
                pxy.append(b)
                csol[i].path_len = a.ctypes.data_as(I32P)
                csol[i].paths_xy = b.ctypes.data_as(I32P)
                csol[i].path_cap = path_cap
        h = ctypes.c_void_p()
        rc = self._lib.mrp_hl_solver_preload(self._h, n_threads, n, cin, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"mrp_hl_solver_preload failed rc={rc}: "
                               f"{self._lib.mrp_hl_solver_last_error(self._h).decode()}")
        return dict(n=n, cin=cin, csol=csol, keep=keep, plen=plen, pxy=pxy, handle=h, want_paths=want_paths,
                    path_cap=path_cap)

    def release(self, prep) -> None:
        if prep.get("handle"):
            self._lib.mrp_hl_preloaded_free(prep["handle"])
            prep["handle"] = None

    def solve_prepared(self, prep, algo: int = ECBS, w: float = 1.3, max_ll_expansions: int = -1,
                       max_hl_expansions: int = -1, mode: int = 0, raw: bool = False):
        """Run the conflict-tree searches of a prepared batch.  raw=True skips building per-instance dicts."""
        opt = mrp_hl_options(algo, w, max_ll_expansions, max_hl_expansions, 0, mode)
        st = mrp_hl_batch_stats()
        rc = self._lib.mrp_hl_solver_solve_preloaded(self._h, ctypes.byref(opt), prep["handle"], prep["csol"],
                                                     ctypes.byref(st))
        if rc != 0:
            raise RuntimeError(f"mrp_hl_solver_solve_preloaded failed rc={rc}: "
                               f"{self._lib.mrp_hl_solver_last_error(self._h).decode()}")
        stats = dict(wall_seconds=st.wall_seconds, rounds=st.rounds, ll_searches=st.ll_searches,
                     ll_expansions=st.ll_expansions, solved=st.solved, build_seconds=st.build_seconds,
                     ll_call_seconds=st.ll_call_seconds, consume_seconds=st.consume_seconds,
                     speculative_searches=st.speculative_searches, wasted_ll_expansions=st.wasted_ll_expansions,
                     root_solved=st.root_solved)
        return (None if raw else self.results_of(prep)), stats

    def solve_stream(self, preps, algo: int = ECBS, w: float = 1.3, max_ll_expansions: int = -1,
                     max_hl_expansions: int = -1):
        """Several prepared batches as ONE stream (mrp_hl.h mrp_hl_solver_solve_stream): no barrier between batches, so
        the dependent chains that end one batch overlap with the next batch's root searches.  Every batch's results are
        those of its own solve_prepared call; read them with result_arrays / results_of.  Returns the stream's statistics."""
        n = len(preps)
        if n == 0:
            raise ValueError("solve_stream needs at least one prepared batch")
        opt = mrp_hl_options(algo, w, max_ll_expansions, max_hl_expansions, 0, 0)
        st = mrp_hl_batch_stats()
        handles = (ctypes.c_void_p * n)(*[p["handle"] for p in preps])
        sols = (ctypes.c_void_p * n)(*[ctypes.cast(p["csol"], ctypes.c_void_p) for p in preps])
        rc = self._lib.mrp_hl_solver_solve_stream(self._h, ctypes.byref(opt), n, handles, sols, ctypes.byref(st))
        if rc != 0:
            raise RuntimeError(f"mrp_hl_solver_solve_stream failed rc={rc}: "
                               f"{self._lib.mrp_hl_solver_last_error(self._h).decode()}")
        return dict(wall_seconds=st.wall_seconds, rounds=st.rounds, ll_searches=st.ll_searches,
                    ll_expansions=st.ll_expansions, solved=st.solved, build_seconds=st.build_seconds,
                    ll_call_seconds=st.ll_call_seconds, consume_seconds=st.consume_seconds,
                    speculative_searches=st.speculative_searches, wasted_ll_expansions=st.wasted_ll_expansions,
                    root_solved=st.root_solved, batches=n)

    def result_arrays(self, prep) -> Dict[str, np.ndarray]:
        """The results of a prepared batch as numpy arrays (status, cost, makespan, hl_expanded, ll_expanded, ll_searches,
        schedule_digest; path_len [n][agents] when the batch delivers schedules through the bulk form)."""
        sv = np.frombuffer(prep["csol"], dtype=_SOLUTION_DTYPE, count=prep["n"])
        out = dict(status=sv["status"].copy(), cost=sv["cost"].copy(), makespan=sv["makespan"].copy(),
                   hl_expanded=sv["high_level_expanded"].copy(), ll_expanded=sv["low_level_expanded"].copy(),
                   ll_searches=sv["n_ll_searches"].copy(), schedule_digest=sv["schedule_digest"].copy())
        if prep["want_paths"] and isinstance(prep["plen"], np.ndarray):
            out["path_len"] = prep["plen"]
        elif prep["want_paths"] and prep["n"]:
            out["path_len"] = np.concatenate([np.asarray(a).ravel() for a in prep["plen"]])
        return out

    def results_of(self, prep) -> List[Dict]:
        out = []
        csol, plen, pxy = prep["csol"], prep["plen"], prep["pxy"]
        for i in range(prep["n"]):
            s = csol[i]
            rec = dict(status=s.status, cost=s.cost, makespan=s.makespan, hl_expanded=s.high_level_expanded,
                       ll_expanded=s.low_level_expanded, ll_searches=s.n_ll_searches, schedule_digest=s.schedule_digest)
            if prep["want_paths"] and s.status == SOLVED:
                assert int(plen[i].max(initial=0)) <= prep["path_cap"]
                rec["paths"] = [pxy[i][a, :plen[i][a]].tolist() for a in range(len(plen[i]))]
            out.append(rec)
        return out

    def solve(self, instances: Sequence[Dict], algo: int = ECBS, w: float = 1.3, max_ll_expansions: int = -1,
              max_hl_expansions: int = -1, n_threads: int = 0, want_paths: bool = True, path_cap: int = 512,
              mode: int = 0):
        prep = self.prepare(instances, n_threads=n_threads, want_paths=want_paths, path_cap=path_cap)
        try:
            return self.solve_prepared(prep, algo=algo, w=w, max_ll_expansions=max_ll_expansions,
                                       max_hl_expansions=max_hl_expansions, mode=mode)
        finally:
            self.release(prep)

    def prioritized_sipp(self, instances: Sequence[Dict], state_cap: int = 512, want_schedules: bool = True):
        """example/mapf_prioritized_sipp.cpp for a batch of instances: every round plans the next agent of all of them.
        want_schedules=False leaves the per-agent state lists out of the returned dicts (benchmarks)."""
        n = len(instances)
        cin = (mrp_hl_instance * max(n, 1))()
        csol = (mrp_hl_sipp_solution * max(n, 1))()
        keep, bufs = [], []
        for i, inst in enumerate(instances):
            ob = np.ascontiguousarray(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
            st = np.ascontiguousarray(np.asarray(inst["starts"], dtype=np.int32).reshape(-1, 2))
            go = np.ascontiguousarray(np.asarray(inst["goals"], dtype=np.int32).reshape(-1, 2))
            keep.append((ob, st, go))
            c = cin[i]
            c.dimx, c.dimy = inst["dimx"], inst["dimy"]
            c.n_obstacles, c.obstacles_xy = len(ob), ob.ctypes.data_as(I32P)
            c.n_agents, c.starts_xy, c.goals_xy = len(st), st.ctypes.data_as(I32P), go.ctypes.data_as(I32P)
            pl = np.zeros(len(st), dtype=np.int32)
            ns = np.zeros(len(st), dtype=np.int32)
            sx = np.zeros((len(st), state_cap, 3), dtype=np.int32)
            bufs.append((pl, ns, sx))
            csol[i].planned, csol[i].n_states = pl.ctypes.data_as(I32P), ns.ctypes.data_as(I32P)
            csol[i].states_xyt, csol[i].state_cap = sx.ctypes.data_as(I32P), state_cap
        st = mrp_hl_batch_stats()
        rc = self._lib.mrp_hl_solver_prioritized_sipp(self._h, n, cin, csol, ctypes.byref(st))
        if rc != 0:
            raise RuntimeError(f"mrp_hl_solver_prioritized_sipp failed rc={rc}: "
                               f"{self._lib.mrp_hl_solver_last_error(self._h).decode()}")
        out = []
        for i in range(n):
            pl, ns, sx = bufs[i]
            out.append(dict(cost=csol[i].cost, expanded=csol[i].low_level_expanded, n_planned=csol[i].n_planned,
                            status=csol[i].status,
                            planned=pl.tolist(),
                            schedules=[sx[a, :ns[a]].tolist() for a in range(len(pl))] if want_schedules else None))
        stats = dict(wall_seconds=st.wall_seconds, rounds=st.rounds, ll_searches=st.ll_searches,
                     ll_expansions=st.ll_expansions, solved=st.solved)
        return out, stats

    def ll_stats(self, reset: bool = False) -> dict:
        st = _ll.mrp_ll_stats()
        self._lib.mrp_hl_solver_ll_stats(self._h, ctypes.byref(st), 1 if reset else 0)
        return {k: (list(getattr(st, k)) if k == "prof" else getattr(st, k)) for k, _ in _ll.mrp_ll_stats._fields_}


class ConflictTree:
    """One conflict tree stepped by the caller (mrp_hl_ct_* of include/mrp_hl.h): `requests()` lists the low-level
    searches that may run now, `deliver(group, results)` feeds back the answers of one group.  The caller decides where
    each search runs (see ct_sharded.py)."""

    def __init__(self, inst: Dict, algo: int = ECBS, w: float = 1.3, map_id: int = 0, spec_width: int = 1,
                 max_ll_expansions: int = -1, max_hl_expansions: int = -1, _lib_path: Optional[str] = None):
        self._lib = load_library(_lib_path)
        self._keep = tuple(np.ascontiguousarray(np.asarray(inst[k], dtype=np.int32).reshape(-1, 2))
                           for k in ("obstacles", "starts", "goals"))
        ob, st, go = self._keep
        cin = mrp_hl_instance(inst["dimx"], inst["dimy"], len(ob), ob.ctypes.data_as(I32P), len(st),
                              st.ctypes.data_as(I32P), go.ctypes.data_as(I32P))
        opt = mrp_hl_options(algo, w, max_ll_expansions, max_hl_expansions, 0, 0)
        h = ctypes.c_void_p()
        rc = self._lib.mrp_hl_ct_create(ctypes.byref(cin), ctypes.byref(opt), map_id, spec_width, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("mrp_hl_ct_create failed rc=%d" % rc)
        self._h = h
        self.n_agents = len(st)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mrp_hl_ct_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def done(self) -> bool:
        return bool(self._lib.mrp_hl_ct_done(self._h))

    def requests(self) -> List[Dict]:
        """Pending searches as plain dicts (group, slot, algo, w, agent, start, goal, vertex_constraints,
        edge_constraints, ctx_paths, max_expansions), in the machine's order (groups are consecutive)."""
        out = []
        job = _ll.mrp_ll_job()
        g, sl = ctypes.c_int32(0), ctypes.c_int32(0)
        for k in range(self._lib.mrp_hl_ct_n_requests(self._h)):
            rc = self._lib.mrp_hl_ct_request(self._h, k, ctypes.byref(job), ctypes.byref(g), ctypes.byref(sl))
            assert rc == 0
            vc = np.ctypeslib.as_array(job.vertex_constraints, (job.n_vertex_constraints, 3)).tolist() \
                if job.n_vertex_constraints else []
            ec = np.ctypeslib.as_array(job.edge_constraints, (job.n_edge_constraints, 5)).tolist() \
                if job.n_edge_constraints else []
            ctx = []
            for a in range(job.n_agents):
                n = job.path_len[a]
                ctx.append(np.ctypeslib.as_array(job.path_xy[a], (n, 2)).tolist() if n else [])
            out.append(dict(group=g.value, slot=sl.value, algo=job.algo, w=job.w, agent=job.agent_idx,
                            start=[job.start_x, job.start_y], goal=[job.goal_x, job.goal_y], vertex_constraints=vc,
                            edge_constraints=ec, ctx_paths=ctx, max_expansions=job.max_expansions))
        return out

    def deliver(self, group: int, results: Sequence[Dict]) -> None:
        """results: per slot dict(status, cost, fmin, expanded, states=[[x, y], ...])."""
        n = len(results)
        cres = (_ll.mrp_ll_result * max(n, 1))()
        keep = []
        for i, r in enumerate(results):
            st = np.zeros((max(len(r["states"]), 1), 3), dtype=np.int32)
            for t, (x, y) in enumerate(r["states"]):
                st[t] = (t, x, y)
            keep.append(st)
            cres[i].status, cres[i].cost, cres[i].fmin = r["status"], r["cost"], r["fmin"]
            cres[i].expanded, cres[i].n_states = r["expanded"], len(r["states"])
            cres[i].states_txy, cres[i].states_cap = st.ctypes.data_as(I32P), len(st)
        rc = self._lib.mrp_hl_ct_deliver(self._h, group, n, cres)
        if rc != 0:
            raise RuntimeError("mrp_hl_ct_deliver(group=%d, n=%d) failed rc=%d" % (group, n, rc))

    def n_requests(self) -> int:
        return int(self._lib.mrp_hl_ct_n_requests(self._h))

    def round_mine(self, ll_handle, rank: int, world: int, rows: np.ndarray, rows_per_rank: np.ndarray) -> int:
        """mrp_hl_ct_round_mine: run this rank's share of the pending searches on the engine `ll_handle` (a mrp_ll_ctx of
        the same library) and pack the results into `rows` ([cap][8 + max_states] int32).  Returns the rows written, or a
        negative MRP_LL_E_* (then rows[0] is the failure row that still has to take part in the all-gather)."""
        fn = self._lib.mrp_hl_ct_round_mine
        fn.restype = ctypes.c_int32
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, I32P, ctypes.c_int32, I32P]
        return int(fn(self._h, ll_handle, rank, world, rows.shape[1] - 8, rows.ctypes.data_as(I32P), rows.shape[0],
                      rows_per_rank.ctypes.data_as(I32P)))

    def deliver_rows(self, gathered: np.ndarray) -> None:
        """mrp_hl_ct_deliver_rows: gathered = [world][rows][8 + max_states] int32, what the all-gather of every rank's
        round_mine rows returned."""
        fn = self._lib.mrp_hl_ct_deliver_rows
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, I32P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        g = np.ascontiguousarray(gathered, dtype=np.int32)
        rc = fn(self._h, g.ctypes.data_as(I32P), g.shape[0], g.shape[1], g.shape[2] - 8)
        if rc != 0:
            raise RuntimeError("mrp_hl_ct_deliver_rows failed rc=%d (a rank reported a failed round, or the rows do not "
                               "match the pending requests)" % rc)

    def solution(self, path_cap: int = 1024) -> Dict:
        sol = mrp_hl_solution()
        plen = np.zeros(max(self.n_agents, 1), dtype=np.int32)
        pxy = np.zeros((max(self.n_agents, 1), path_cap, 2), dtype=np.int32)
        sol.path_len, sol.paths_xy, sol.path_cap = plen.ctypes.data_as(I32P), pxy.ctypes.data_as(I32P), path_cap
        rc = self._lib.mrp_hl_ct_solution(self._h, ctypes.byref(sol))
        if rc != 0:
            raise RuntimeError("mrp_hl_ct_solution failed rc=%d (not done yet?)" % rc)
        rec = dict(status=sol.status, cost=sol.cost, makespan=sol.makespan, hl_expanded=sol.high_level_expanded,
                   ll_expanded=sol.low_level_expanded, ll_searches=sol.n_ll_searches)
        if sol.status == SOLVED:
            rec["paths"] = [pxy[a, :plen[a]].tolist() for a in range(self.n_agents)]
        return rec
