#!/usr/bin/env python3
"""Benchmark of the hot path: ECBS (w = 1.3) low-level searches on 32x32_obst204-shaped instances, batched on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one batch of B fresh synthetic instances per GPU solved to completion by the host conflict-tree driver, every
low-level search of which runs in the HIP kernel (through the C-ABI).  metric = low-level node expansions per second
(the reference's `lowLevelExpanded` counter, example/ecbs.cpp:476-479,599) over the whole job; `instances_per_s` rides
along.  Instances are generated (splitmix64, seeds 1000*agents + k) before the timed region and their maps are uploaded
to HBM before it starts, as the reference constructs its Environment before its Timer (example/ecbs.cpp:576-582).

Extra objects:
  roofline     — dominant kernel mrp_ll_persistent_kernel (session mode: one resident launch per host thread per step):
                 achieved = 128 B/expansion x expansions of the timed launches / sum of their hipEvent
                 durations; bound = HBM (8 TB/s); traffic = HBM bytes per launch scaled from the committed PMC passes.
                 The fraction is tiny by construction — the path is an issue-bound replay of sequential heaps whose
                 working set sits in LDS / L2 — so `issue_bound` rides along: the chip-wide ceiling of
                 the kernel's measured instruction stream (instructions per expansion from the committed PMC passes x 4
                 cycles per issue slot, 256 CUs x 4 SIMDs) and the fraction of it this run reached.
  cpu_baseline — the oracle's CPU restatement (kind "port"; the reference needs Boost/yaml-cpp and cannot be built here)
                 timed single-threaded on a bounded sample of the same workload, on this box's host cores, with a
                 per-instance parity check against the GPU results; cpu_baseline_all_cores repeats the sample with one
                 instance per thread on all host threads the bench uses (SURVEY.md §8d asks for both).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# One HIP stream per host worker thread: give each its own hardware queue (the ROCm default of 4 makes streams that
# share a queue serialise their kernels — fatal for resident kernels, which only end when their host thread is done;
# 24 > 16 worker streams + torch's own).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ALGO_BYTES_PER_EXPANSION = 128
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=65536, help="instances per GPU per step")
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--threads", type=int, default=0, help="host worker threads per GPU (0 = auto)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--lds-nodes", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="instances of step 0 timed on the CPU oracle (rank 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-ll-expansions", type=int, default=50000,
                    help="harness cap per instance (the reference has none and never returns on infeasible inputs); "
                         "applied identically to the GPU path and to the CPU baseline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}", file=sys.stderr)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # MRP_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo — only to rehearse the multi-rank control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); never used for a reported number
    rehearsal = os.environ.get("MRP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)
        dist = dist_mod

    from libmultirobotplanning_amd import hl
    hc = os.cpu_count() or 8
    threads = args.threads or max(2, min(16, hc // max(1, world)))
    solver = hl.BatchSolver(device=local_rank, n_threads=threads, slots=args.slots, lds_nodes=args.lds_nodes)

    B, K, W = args.instances, args.steps, args.warmup

    from libmultirobotplanning_amd import sharding

    def batch(step_idx):
        # distinct seeds per (rank, step); step indices of the timed steps start after the warm-up ones
        base = sharding.seed_base(args.agents, rank, K + W, step_idx, B)
        return [hl.generate_instance(base + k, 32, 32, 204, args.agents) for k in range(B)]

    batches = [batch(i) for i in range(K + W)]
    # marshal the batches and upload their static maps to HBM before the timed region (the reference constructs its
    # Environment before it starts its Timer, example/ecbs.cpp:576-582); the timed region is the searches only
    prepared = [solver.prepare(b, want_paths=False) for b in batches]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        solver.solve_prepared(prepared[i], algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions, raw=True)
    solver.ll_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    exp_total = 0
    solved_total = 0
    searches_total = 0
    for i in range(W, W + K):
        _, st = solver.solve_prepared(prepared[i], algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions,
                                      raw=True)
        exp_total += st["ll_expansions"]
        solved_total += st["solved"]
        searches_total += st["ll_searches"]
    barrier()
    elapsed = time.perf_counter() - t0
    lls = solver.ll_stats()
    first_results = solver.results_of(prepared[W]) if K > 0 else []

    # totals: the only exchange this path needs (max of the elapsed times, sum of the counters)
    elapsed_max, (exp_all, solved_all, searches_all, inst_all) = sharding.reduce_totals(
        dist, "cpu" if rehearsal else "cuda", elapsed, [exp_total, solved_total, searches_total, K * B])

    if rank == 0:
        kernel_s = lls["kernel_ms"] / 1e3
        achieved = ALGO_BYTES_PER_EXPANSION * lls["expansions"] / max(kernel_s, 1e-12) / 1e9
        # HBM bytes per launch: bench.py cannot run rocprofv3's PMC passes on itself, so it scales the bytes per expansion
        # measured by the committed passes over this same program (profiles/hbm_traffic_pmc.json, FETCH_SIZE + WRITE_SIZE
        # in separate passes, raw counter values) by the expansions of its own launches; null when the file is absent
        traffic = None
        issue_bound = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")) as f:
                pmc = json.load(f)
            traffic = pmc["bytes_per_expansion_raw"] * lls["expansions"] / max(lls["launches"], 1)
            ipe = pmc["instructions_per_expansion"]
            ceiling = 256 * 4 * pmc.get("shader_clock_hz", 2.2e9) / (4.0 * ipe)
            issue_bound = {"instructions_per_expansion": ipe, "ceiling_expansions_per_s": ceiling,
                           "frac": (exp_all / elapsed_max) / world / ceiling,
                           "source": "profiles/hbm_traffic_pmc.json (SQ_INSTS_* passes, scripts/pmc_ll.sh)"}
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "low_level_node_expansions_per_sec",
            "value": exp_all / elapsed_max,
            "unit": "expansions/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed_max / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "ECBS w=1.3, synthetic 32x32_obst204-shaped instances, agents%d, %d instances/GPU/step "
                                   "(configs[1] shape)" % (args.agents, B),
                       "instances_per_gpu_per_step": B, "agents": args.agents, "host_threads_per_gpu": threads,
                       "max_ll_expansions_per_instance": args.max_ll_expansions,
                       "parallelism": "instances sharded per GPU, no data-path collective"},
            "instances_per_s": inst_all / elapsed_max,
            "solved": int(solved_all),
            "instances": int(inst_all),
            "ll_searches": int(searches_all),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "achieved_all_launches_concurrently": ALGO_BYTES_PER_EXPANSION * (exp_all / elapsed_max) / 1e9,
                "kernel": "mrp_ll_persistent_kernel",
                "launches": lls["launches"],
                "avg_launch_ms": lls["kernel_ms"] / max(lls["launches"], 1),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EXPANSION * lls["expansions"] / max(lls["launches"], 1),
                "issue_bound": issue_bound,
                "note": "rank-0 launches of the timed region: session mode keeps one resident launch per host thread "
                        "per step (fed through the pinned-host job ring), and the launches of the threads overlap in time",
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU legs run at N=1 only (rank 0)
            import oracle
            oracle.build()
            n = min(args.cpu_sample, B)
            t_cpu = 0.0
            e_cpu = 0
            mism = 0
            for inst, r in zip(batches[W][:n], first_results[:n]):
                o = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=args.max_ll_expansions, path_cap=1024)
                t_cpu += o["elapsed_ns"] / 1e9
                e_cpu += o["ll_expanded"]
                if o["rc"] == 1:
                    if (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) != (
                            hl.SOLVED, o["cost"], o["makespan"], o["hl_expanded"], o["ll_expanded"]):
                        mism += 1
                elif r["status"] != hl.CAP:  # capped on the CPU => must be capped on the GPU too
                    mism += 1
            out["cpu_baseline"] = {
                "value": e_cpu / max(t_cpu, 1e-12), "unit": "expansions/s", "cores": 1, "kind": "port",
                "sample": "first %d instances of timed step 0 (rank 0), oracle ECBS w=1.3, g++ -O3, search() time only"
                          % n,
                "instances_per_s": n / max(t_cpu, 1e-12), "seconds": t_cpu,
                "host_cpus": hc, "parity_mismatches_vs_gpu": mism,
            }
            # SURVEY.md §8(d) also asks for "one instance per thread on all host cores": the same sample again on a
            # thread pool (the oracle call is a ctypes call, i.e. runs without the GIL); wall time of the whole pool
            from concurrent.futures import ThreadPoolExecutor
            pool_n = max(1, min(threads, hc))
            t1 = time.perf_counter()
            with ThreadPoolExecutor(pool_n) as ex:
                outs = list(ex.map(lambda inst: oracle.mapf_solve(oracle.ECBS, inst, w=1.3,
                                                                  cap_total=args.max_ll_expansions, path_cap=1024),
                                   batches[W][:n]))
            wall = time.perf_counter() - t1
            busy = sum(o["elapsed_ns"] for o in outs) / 1e9  # search() time summed over the concurrent searches
            out["cpu_baseline_all_cores"] = {
                "value": sum(o["ll_expanded"] for o in outs) / max(busy / pool_n, 1e-12), "unit": "expansions/s",
                "cores": pool_n, "kind": "port", "pool_wall_seconds": wall, "search_seconds_sum": busy,
                "sample": "the same %d instances, one instance per thread on %d threads; expansions / (sum of search() "
                          "times / threads), i.e. Python marshalling between searches is not charged" % (n, pool_n),
            }
        print(json.dumps(out), flush=True)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
