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
res, st = solver.solve(insts, algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=50000)
elapsed, sums = sharding.reduce_totals(dist, "cpu", 1.0 + rank, [st["ll_expansions"], st["solved"], B])
# strong-scaling split of one fixed list
idx = sharding.shard_indices(10, rank, world)
if rank == 0:
    print(json.dumps(dict(elapsed=elapsed, sums=sums, idx=idx)))
dist.destroy_process_group()
"""


@pytest.mark.timeout(300)
def test_two_ranks_gloo(oracle_mod, tmp_path):
    from libmultirobotplanning_amd import hl, sharding
    import test_host_drivers_cpu as hd  # builds the CPU driver library
    os.makedirs(hd.BUILD, exist_ok=True)
    lib = os.path.join(hd.BUILD, "libmrp_hl_cpu.so")
    if not os.path.exists(lib):
        srcs = [os.path.join(ROOT, "libmultirobotplanning_amd", "csrc", "hl", "mrp_hl.cpp"),
                os.path.join(ROOT, "tests", "support", "mock_ll.cpp")]
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-I",
                               os.path.join(ROOT, "include"), "-o", lib] + srcs +
                              ["-L", os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
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
