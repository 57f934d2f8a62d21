"""N > 1 path on CPU: two ranks (gloo) each solve their shard of the instances through the host drivers (oracle-backed
mock low level, see tests/support/mock_ll.cpp) and reduce totals exactly as bench.py does on GPUs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
from libmultirobotplanning_amd import hl, sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
solver = hl.BatchSolver(device=0, n_threads=2, _lib_path={lib!r})
B = 12
# weak-scaling shard: every rank has its own seeds
base = sharding.seed_base(10, rank, 1, 0, B)
insts = [hl.generate_instance(base + k, 32, 32, 204, 10) for k in range(B)]
# ... and solves them as bench.py does: its batches as one stream (mrp_hl_solver_solve_stream)
preps = [solver.prepare(insts[:7], want_paths=False), solver.prepare(insts[7:], want_paths=False)]
st = solver.solve_stream(preps, algo=hl.ECBS, w=1.3, max_ll_expansions=50000)
for p in preps:
    solver.release(p)
elapsed, sums = sharding.reduce_totals(dist, "cpu", 1.0 + rank, [st["ll_expansions"], st["solved"], B])
# strong-scaling split of one fixed list
idx = sharding.shard_indices(10, rank, world)
if rank == 0:
    print(json.dumps(dict(elapsed=elapsed, sums=sums, idx=idx)))
dist.destroy_process_group()
"""


def _cpu_driver_lib():
    """The host drivers built against the oracle-backed mock low level (tests/support/mock_ll.cpp); rebuilt when stale."""
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    lib = os.path.join(build, "libmrp_hl_cpu.so")
    srcs = [os.path.join(ROOT, "libmultirobotplanning_amd", "csrc", "hl", "mrp_hl.cpp"),
            os.path.join(ROOT, "tests", "support", "mock_ll.cpp")]
    deps = srcs + [os.path.join(ROOT, "libmultirobotplanning_amd", "csrc", "hl", f) for f in
                   ("ct_solver.hpp", "grid_mapf.hpp", "exact_heap.hpp", "grid2d_astar.hpp", "instance_io.hpp")] + \
        [os.path.join(ROOT, "include", "mrp_hl.h"), os.path.join(ROOT, "include", "mrp_ll.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-I",
                               os.path.join(ROOT, "include"), "-o", lib] + srcs +
                              ["-L", os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return lib


@pytest.mark.timeout(300)
def test_two_ranks_gloo(oracle_mod, tmp_path):
    from libmultirobotplanning_amd import hl, sharding
    lib = _cpu_driver_lib()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, lib=lib))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    # reference: the same 24 instances solved in one process by the oracle
    exp = 0
    for rank in range(2):
        base = sharding.seed_base(10, rank, 1, 0, 12)
        for k in range(12):
            o = oracle_mod.mapf_solve(oracle_mod.ECBS, hl.generate_instance(base + k, 32, 32, 204, 10), w=1.3,
                                      cap_total=50000)
            exp += o["ll_expanded"]
    assert got["sums"][0] == exp
    assert got["sums"][2] == 24
    assert got["elapsed"] == 2.0          # max over ranks
    assert got["idx"] == [0, 2, 4, 6, 8]
    assert sharding.shard_indices(10, 1, 2) == [1, 3, 5, 7, 9]


CT_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
import oracle                      # TEST executor only: the product executor is ct_sharded.gpu_executor (one MI355X per rank)
from libmultirobotplanning_amd import hl, ct_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
cases = json.load(open({cases!r}))
out = []
for case in cases:
    # rank 0 owns the input; the static map and the agents reach the other rank by broadcast
    inst = ct_sharded.broadcast_instance(case["inst"] if rank == 0 else None, dist, "cpu")
    m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
    def run(reqs):
        res = []
        for r in reqs:
            o = oracle.ll_search(r["algo"], m, r["agent"], r["start"], r["goal"], r["vertex_constraints"],
                                 r["edge_constraints"], r["ctx_paths"], w=r["w"], cap_expansions=r["max_expansions"])
            res.append(dict(status=2 if o["rc"] == -1 else (0 if o["success"] else 1), cost=o["cost"], fmin=o["fmin"],
                            expanded=o["expanded"], states=[s[1:] for s in o["states"]]))
        return res
    r = ct_sharded.solve_sharded(inst, run, dist, algo=case["algo"], w=1.3, spec_width=case["spec"], device="cpu",
                                 max_ll_expansions=case.get("cap", -1), _lib_path={lib!r})
    # ... and the product's round: native pack / deliver (mrp_hl_ct_round_mine / _deliver_rows) on an engine handle — here
    # the oracle-backed C-ABI inside the CPU build of the drivers (tests/support/mock_ll.cpp)
    import ctypes
    eng = ct_sharded.NativeEngine(inst, lib=ctypes.CDLL({lib!r}))
    rn = ct_sharded.solve_sharded(inst, eng, dist, algo=case["algo"], w=1.3, spec_width=case["spec"], device="cpu",
                                  max_ll_expansions=case.get("cap", -1), _lib_path={lib!r})
    eng.close()
    r["native_same"] = all(r[k] == rn[k] for k in ("status", "cost", "makespan", "hl_expanded", "ll_expanded")) and \
        r.get("paths") == rn.get("paths") and rn["searches_run_here"] > 0
    out.append(r)
# a rank whose share of a round fails must not leave the other rank blocked in the round's all-gather: BOTH raise
inst = ct_sharded.broadcast_instance(cases[2]["inst"] if rank == 0 else None, dist, "cpu")
calls = [0]
def failing(reqs):
    calls[0] += 1
    if rank == 1 and calls[0] == 3:
        raise ValueError("executor broke on rank 1")
    return run(reqs)
m = dict(dimx=inst["dimx"], dimy=inst["dimy"], obstacles=inst["obstacles"])
try:
    ct_sharded.solve_sharded(inst, failing, dist, algo=cases[2]["algo"], w=1.3, spec_width=2, device="cpu", _lib_path={lib!r})
    raised = ""
except RuntimeError as e:
    raised = str(e)
out.append(dict(raised=raised))
with open({outdir!r} + "/rank%d.json" % rank, "w") as f:  # (two ranks printing to one pipe can interleave)
    json.dump(out, f)
dist.destroy_process_group()
"""


@pytest.mark.timeout(600)
def test_one_conflict_tree_sharded_over_two_ranks(oracle_mod, bench_instances, oracle_expected, tmp_path):
    """SURVEY.md §8e / north_star: ONE heavy instance, the searches of every round split over two ranks (gloo here, RCCL
    on GPUs): broadcast of the instance, one all-gather of the results per round, children committed in the reference's
    pop order.  Both ranks must end with the single-rank result — (cost, makespan, HL, LL, paths) of the oracle — and both
    must have run searches."""
    from libmultirobotplanning_amd import hl
    lib = _cpu_driver_lib()
    picks = [("map_32by32_obst204_agents50_ex1", hl.ECBS, "ecbs_w1.3", 2), ("map_32by32_obst204_agents30_ex2", hl.ECBS, "ecbs_w1.3", 4),
             ("map_8by8_obst12_agents6_ex1", hl.CBS, "cbs", 2), ("map_8by8_obst12_agents8_ex3", hl.CBS, "cbs", 4)]
    cases = [dict(inst=bench_instances[n], algo=a, spec=k) for n, a, _, k in picks]
    cases.append(dict(inst=bench_instances["map_8by8_obst12_agents8_ex0"], algo=hl.CBS, spec=2, cap=20000))  # capped
    cfile = tmp_path / "cases.json"
    cfile.write_text(json.dumps(cases))
    script = tmp_path / "ct_worker.py"
    script.write_text(CT_WORKER.format(root=ROOT, lib=lib, cases=str(cfile), outdir=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29579", str(script)],
                         env=env, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    per_rank = {}
    for rank in (0, 1):
        with open(tmp_path / ("rank%d.json" % rank)) as f:
            per_rank[rank] = json.load(f)
    for i, (n, _, key, _) in enumerate(picks):
        e = oracle_expected[n][key]
        for rank in (0, 1):
            r = per_rank[rank][i]
            assert (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
                hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"]), (n, rank)
            assert r["searches_run_here"] > 0
            assert r["native_same"], (n, rank)
        assert per_rank[0][i]["paths"] == per_rank[1][i]["paths"]
        assert per_rank[0][i]["rounds"] == per_rank[1][i]["rounds"]
        # the two ranks split the work: together they ran every consumed search (plus any look-ahead that was not)
        assert per_rank[0][i]["searches_run_here"] + per_rank[1][i]["searches_run_here"] >= per_rank[0][i]["ll_searches"]
        assert per_rank[0][i]["rounds"] < per_rank[0][i]["ll_searches"]
    assert per_rank[0][-2]["status"] == hl.CAP and per_rank[1][-2]["status"] == hl.CAP
    assert per_rank[0][-2]["native_same"] and per_rank[1][-2]["native_same"]
    # the failure case: both ranks raised, naming the rank that failed
    assert "rank(s) [1]" in per_rank[0][-1]["raised"] and "rank(s) [1]" in per_rank[1][-1]["raised"]
