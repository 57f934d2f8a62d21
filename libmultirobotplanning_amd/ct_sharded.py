"""ONE conflict tree whose low-level searches are sharded over the ranks of a torch.distributed job (SURVEY.md §8e,
BASELINE.json north_star: "conflict-tree node batches shard one-per-GPU over RCCL/xGMI: broadcast of the static map,
all-gather of incumbent costs").

How it stays exact.  Every rank runs the same deterministic conflict-tree machine (hl.ConflictTree = the C++ state
machine of csrc/hl/ct_solver.hpp: CBS::search cbs.hpp:85-172 / ECBS::search ecbs.hpp:109-288 cut at the low-level
calls) with the same look-ahead width, so all ranks see the same list of pending request groups each round — the two
children of the node that was popped plus, with spec_width > 1, the children of the nodes the loop will pop next.  Group
j of a round is searched by rank j % world on its own GPU; one all-gather per round hands every rank every result, and
all of them deliver the groups in the same order.  Children are committed strictly in the reference's pop order
(ct_solver.hpp), so cost, makespan, highLevelExpanded, lowLevelExpanded and the paths equal the single-rank run's.

Collectives (RCCL when the backend is "nccl", gloo on CPU): one broadcast of the instance (the static map, starts,
goals) from rank 0, then one fixed-shape int32 all-gather per round — per result: group, slot, status, cost (the
incumbent's contribution), fmin, expansions and the path.  Messages are a few KB: latency-bound, as SURVEY §8e expects.

A round is native: `mrp_hl_ct_round_mine` (libmrp_hl) runs this rank's searches straight from the tree's own job structs
on the rank's engine and packs the rows, `mrp_hl_ct_deliver_rows` unpacks all ranks' rows and steps the tree; Python only
moves one int32 buffer through the collective.  A failing rank contributes a failure row, so all ranks raise together
instead of leaving the others blocked in the all-gather.  (`executor` callables — Python per request — remain for tests.)
"""
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import hl

_FAILED_ROUND = -(2 ** 31)  # group id of a failure row (mrp_hl.cpp kFailedRound): no conflict-tree group uses it
_HDR = 8  # int32 words in front of the path: group, slot, status, cost, fmin, expanded lo, expanded hi, n_states


def broadcast_instance(inst: Optional[Dict], dist, device: str = "cpu", src: int = 0) -> Dict:
    """Rank `src` holds the instance; everybody returns the same dict (the "broadcast of the static map")."""
    import torch
    if dist is None or dist.get_world_size() == 1:
        return inst
    me = dist.get_rank()
    hdr = torch.zeros(4, dtype=torch.int32, device=device)
    if me == src:
        hdr[:] = torch.tensor([inst["dimx"], inst["dimy"], len(inst["obstacles"]), len(inst["starts"])], dtype=torch.int32)
    dist.broadcast(hdr, src=src)
    dimx, dimy, n_ob, n_ag = (int(v) for v in hdr.tolist())
    body = torch.zeros((n_ob + 2 * n_ag) * 2, dtype=torch.int32, device=device)
    if me == src:
        flat = [c for o in inst["obstacles"] for c in o] + [c for s in inst["starts"] for c in s] + \
               [c for g in inst["goals"] for c in g]
        body[:] = torch.tensor(flat, dtype=torch.int32)
    dist.broadcast(body, src=src)
    v = body.cpu().numpy().reshape(-1, 2)
    return dict(dimx=dimx, dimy=dimy, obstacles=v[:n_ob].tolist(), starts=v[n_ob:n_ob + n_ag].tolist(),
                goals=v[n_ob + n_ag:].tolist())


class NativeEngine:
    """The product executor: this rank's engine as a mrp_ll_ctx handle for the native rounds (mrp_hl_ct_round_mine).
    `lib` = the ctypes library that exports the mrp_ll_* C-ABI: libmrp_ll.so on a GPU (no CPU fallback); tests pass the
    oracle-backed CPU build (tests/support/mock_ll.cpp inside tests/_build/libmrp_hl_cpu.so)."""

    def __init__(self, inst: Dict, device: int = 0, max_horizon: int = 512, lib=None):
        from . import ll
        import ctypes
        self._lib = lib if lib is not None else ll.load_library()
        opt = ll.mrp_ll_options(device, 1, 64, 0, max_horizon, 0, 0, 0)
        h = ctypes.c_void_p()
        self._lib.mrp_ll_create.restype = ctypes.c_int
        self._lib.mrp_ll_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        if self._lib.mrp_ll_create(ctypes.byref(opt), ctypes.byref(h)) != 0:
            raise RuntimeError("mrp_ll_create failed (no HIP device?)")
        self.handle = h
        ob = np.ascontiguousarray(np.asarray(inst["obstacles"], dtype=np.int32).reshape(-1, 2))
        mid = ctypes.c_int32(-1)
        self._lib.mrp_ll_upload_map.restype = ctypes.c_int
        self._lib.mrp_ll_upload_map.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        rc = self._lib.mrp_ll_upload_map(h, inst["dimx"], inst["dimy"], len(ob), ob.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                         ctypes.byref(mid))
        if rc != 0 or mid.value != 0:
            raise RuntimeError("mrp_ll_upload_map failed rc=%d" % rc)

    def close(self):
        if getattr(self, "handle", None):
            import ctypes
            self._lib.mrp_ll_destroy.restype = None
            self._lib.mrp_ll_destroy.argtypes = [ctypes.c_void_p]
            self._lib.mrp_ll_destroy(self.handle)
            self.handle = None



def gpu_executor(inst: Dict, device: int = 0, max_horizon: int = 512):
    """This rank's MI355X for solve_sharded (native rounds)."""
    return NativeEngine(inst, device=device, max_horizon=max_horizon)


def python_executor(inst: Dict, device: int = 0, max_horizon: int = 512) -> Callable[[Sequence[Dict]], List[Dict]]:
    """The same engine behind a per-request Python callable (round 2's path; kept for comparison)."""
    from . import ll
    eng = ll.LowLevelEngine(device=device, n_tickets=1, slots=64, max_horizon=max_horizon)
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])

    def run(reqs: Sequence[Dict]) -> List[Dict]:
        jobs = [ll.LLJob(map_id=mid, algo=r["algo"], start=r["start"], goal=r["goal"], agent_idx=r["agent"], w=r["w"],
                         vertex_constraints=r["vertex_constraints"], edge_constraints=r["edge_constraints"],
                         ctx_paths=r["ctx_paths"], max_expansions=r["max_expansions"]) for r in reqs]
        out = []
        for r in (eng.search_batch(jobs) if jobs else []):
            out.append(dict(status=r.status, cost=r.cost, fmin=r.fmin, expanded=r.expanded,
                            states=[s[1:] for s in r.states]))
        return out

    run.close = eng.close
    return run


def solve_sharded(inst: Dict, executor: Callable[[Sequence[Dict]], List[Dict]], dist=None, algo: int = hl.ECBS,
                  w: float = 1.3, spec_width: int = 0, max_ll_expansions: int = -1, max_hl_expansions: int = -1,
                  device: str = "cpu", max_states: int = 512, _lib_path: Optional[str] = None) -> Dict:
    """Solve `inst` (identical on all ranks: see broadcast_instance) with the searches of every round sharded over the
    ranks of `dist`.  Returns the same dict on every rank: hl.ConflictTree.solution() + rounds / searches_run_here."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    me = dist.get_rank() if dist is not None else 0
    if spec_width <= 0:
        spec_width = max(1, world)  # one node's children per rank and round
    ct = hl.ConflictTree(inst, algo=algo, w=w, map_id=0, spec_width=spec_width, max_ll_expansions=max_ll_expansions,
                         max_hl_expansions=max_hl_expansions, _lib_path=_lib_path)
    rounds = ran = 0
    native = isinstance(executor, NativeEngine)
    import time
    t_search = t_coll = t_deliver = 0.0
    width = _HDR + max_states
    # Buffers of the native rounds, allocated ONCE: this rank's rows (pinned when they travel to a GPU), their device
    # copy, the gathered rows of all ranks on the device and on the host.  A round moves `stride` rows per rank — the
    # largest share of the round — through views of these; nothing is allocated, stacked or converted per round.
    cap_rows = max(8, 4 * spec_width + 4)
    on_gpu = native and device != "cpu" and world > 1
    if native:
        rows_t = torch.zeros((cap_rows, width), dtype=torch.int32)
        gath_t = torch.zeros((world * cap_rows * width,), dtype=torch.int32)
        if on_gpu:
            rows_t, gath_t = rows_t.pin_memory(), gath_t.pin_memory()
            send_dev = torch.zeros((cap_rows, width), dtype=torch.int32, device=device)
            gath_dev = torch.zeros((world * cap_rows * width,), dtype=torch.int32, device=device)
        rows = rows_t.numpy()
        gath = gath_t.numpy()
        per_rank = np.zeros(world, dtype=np.int32)
    try:
        while not ct.done():
            if native:
                # one native round: this rank's searches + rows; ONE all-gather; all ranks step the tree with the same rows
                n_req = ct.n_requests()
                if n_req == 0:
                    raise RuntimeError("conflict tree is neither done nor asking for searches")
                if n_req > cap_rows:
                    raise RuntimeError("more requests in a round than the row buffers hold")
                t0 = time.perf_counter()
                n_mine = ct.round_mine(executor.handle, me, world, rows[:n_req], per_rank)
                t1 = time.perf_counter()
                stride = max(int(per_rank.max()), 1)
                ran += max(n_mine, 0)
                rounds += 1
                if world == 1:
                    gathered = rows[None, :stride]
                else:
                    n_words = world * stride * width
                    if on_gpu:
                        send_dev[:stride].copy_(rows_t[:stride], non_blocking=True)
                        dist.all_gather_into_tensor(gath_dev[:n_words], send_dev[:stride].reshape(-1))
                        gath_t[:n_words].copy_(gath_dev[:n_words])
                    else:
                        try:
                            dist.all_gather_into_tensor(gath_t[:n_words], rows_t[:stride].reshape(-1))
                        except (RuntimeError, AttributeError):  # a backend without the flat form
                            dist.all_gather([gath_t[k * stride * width:(k + 1) * stride * width] for k in range(world)],
                                            rows_t[:stride].reshape(-1))
                    gathered = gath[:n_words].reshape(world, stride, width)
                t2 = time.perf_counter()
                ct.deliver_rows(gathered)  # raises on EVERY rank if any rank sent a failure row
                t3 = time.perf_counter()
                t_search += t1 - t0
                t_coll += t2 - t1
                t_deliver += t3 - t2
                continue
            reqs = ct.requests()
            if not reqs:
                raise RuntimeError("conflict tree is neither done nor asking for searches")
            groups: List[int] = []
            for r in reqs:
                if not groups or groups[-1] != r["group"]:
                    groups.append(r["group"])
            owner = {g: j % world for j, g in enumerate(groups)}
            mine = [r for r in reqs if owner[r["group"]] == me]
            if world == 1:
                res = executor(mine)
                ran += len(mine)
                rounds += 1
                rows = [(r["group"], r["slot"], x) for r, x in zip(mine, res)]
            else:
                # A rank whose share of the round fails (executor error, a path longer than max_states) still takes part
                # in the round's collective and says so in its first row; every rank then raises together instead of
                # one rank leaving the others blocked in all_gather.
                per_rank = max(1, max(sum(1 for r in reqs if owner[r["group"]] == k) for k in range(world)))
                buf = torch.zeros((per_rank, _HDR + max_states), dtype=torch.int32)
                failure = None
                try:
                    res = executor(mine)
                    for i, (r, x) in enumerate(zip(mine, res)):
                        n = len(x["states"])
                        if n > max_states:
                            raise RuntimeError("path longer than max_states")
                        buf[i, :_HDR] = torch.tensor([r["group"], r["slot"], x["status"], x["cost"], x["fmin"],
                                                      x["expanded"] & 0x7FFFFFFF, x["expanded"] >> 31, n], dtype=torch.int32)
                        if n:
                            buf[i, _HDR:_HDR + n] = torch.tensor([p[0] | (p[1] << 16) for p in x["states"]], dtype=torch.int32)
                except Exception as e:  # noqa: BLE001 — reported to every rank below
                    failure = e
                    buf.zero_()
                    buf[0, 0] = _FAILED_ROUND
                ran += len(mine)
                rounds += 1
                buf = buf.to(device)
                gathered = [torch.zeros_like(buf) for _ in range(world)]
                dist.all_gather(gathered, buf)
                bad = [k for k in range(world) if int(gathered[k][0, 0]) == _FAILED_ROUND]
                if bad:
                    raise RuntimeError("sharded round failed on rank(s) %s" % bad) from failure
                rows = []
                for k in range(world):
                    n_k = sum(1 for r in reqs if owner[r["group"]] == k)
                    t = gathered[k].cpu().numpy()
                    for i in range(n_k):
                        h = t[i, :_HDR]
                        n = int(h[7])
                        cells = t[i, _HDR:_HDR + n]
                        rows.append((int(h[0]), int(h[1]),
                                     dict(status=int(h[2]), cost=int(h[3]), fmin=int(h[4]),
                                          expanded=int(h[5]) | (int(h[6]) << 31),
                                          states=[[int(c) & 0xFFFF, int(c) >> 16] for c in cells])))
            by_group: Dict[int, List] = {}
            for g, slot, x in rows:
                by_group.setdefault(g, []).append((slot, x))
            for g in groups:  # the same order on every rank
                ct.deliver(g, [x for _, x in sorted(by_group[g], key=lambda v: v[0])])
                if ct.done():
                    break
        out = ct.solution()
    finally:
        ct.close()
    out["rounds"] = rounds
    out["searches_run_here"] = ran
    if native and rounds:  # where a round's time goes on this rank (microseconds per round)
        out["us_per_round"] = dict(search_and_pack=1e6 * t_search / rounds, collective=1e6 * t_coll / rounds,
                                   deliver=1e6 * t_deliver / rounds)
    return out
