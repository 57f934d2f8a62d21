#!/usr/bin/env python3
"""The SAME conflict tree at world size 1 and at world size 2 on one GPU (gloo between the ranks: RCCL refuses two ranks on
one device) — does sharding one tree pay for its round trip?  Prints one JSON object per world size on rank 0:
seconds, rounds, and microseconds per round split into search + pack / collective / deliver.

  python scripts/rehearse_sharded.py                       # world 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      scripts/rehearse_sharded.py                         # world 2, both ranks on cuda:0
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")


def main():
    import torch
    import torch.distributed as dist
    from libmultirobotplanning_amd import ct_sharded, hl
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    d = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        d = dist
    names = sys.argv[1:] or ["map_32by32_obst204_agents100_ex0", "map_32by32_obst204_agents100_ex2", "map_32by32_obst204_agents50_ex3"]
    with open(os.path.join(ROOT, "tests", "golden", "bench_instances.json")) as f:
        insts = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_expected.json")) as f:
        gold = json.load(f)
    out = {"world": world, "transport": "gloo, both ranks on one GPU" if world > 1 else "none", "trees": {}}
    for name in names:
        inst = ct_sharded.broadcast_instance(insts[name] if rank == 0 else None, d, "cpu")
        run = ct_sharded.gpu_executor(inst, device=0)
        try:
            for spec in sorted({1, 2, max(2, world)}):
                ct_sharded.solve_sharded(inst, run, d, algo=hl.ECBS, w=1.3, spec_width=spec, device="cpu")  # warm-up
                if d is not None:
                    d.barrier()
                t0 = time.perf_counter()
                r = ct_sharded.solve_sharded(inst, run, d, algo=hl.ECBS, w=1.3, spec_width=spec, device="cpu")
                if d is not None:
                    d.barrier()
                dt = time.perf_counter() - t0
                e = gold[name]["ecbs_w1.3"]
                out["trees"].setdefault(name, {})["spec_width_%d" % spec] = {
                    "seconds": dt, "rounds": r["rounds"], "searches_run_on_rank0": r["searches_run_here"],
                    "us_per_round": r.get("us_per_round"),
                    "matches_golden": (r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (e["cost"], e["makespan"], e["hl"], e["ll"])}
        finally:
            run.close()
    if rank == 0:
        print(json.dumps(out))
    if d is not None:
        d.destroy_process_group()


if __name__ == "__main__":
    main()
