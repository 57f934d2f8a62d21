#!/usr/bin/env python3
"""Benchmark of the hot path: ECBS (w = 1.3) low-level searches on 32x32_obst204-shaped instances, batched on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one batch of B fresh synthetic agents10 instances per GPU (BASELINE.json configs[1] shape) solved to
completion by the host conflict-tree driver, every low-level search of which runs in the HIP kernel (through the C-ABI).
metric = low-level node expansions per second (the reference's `lowLevelExpanded` counter, example/ecbs.cpp:476-479,599)
over the whole job; `instances_per_s` rides along.  Instances are generated natively (splitmix64, seeds 1000*agents + k,
one mrp_hl_generate_instances call per batch) before the timed region and their maps are uploaded to HBM before it
starts, as the reference constructs its Environment before its Timer (example/ecbs.cpp:576-582).

The K timed steps go to the solver as ONE STREAM of K batches (mrp_hl_solver_solve_stream, config.steps_as_one_stream):
every batch is solved completely and delivered inside the timed region, bracketed by the barrier + synchronize pair, but
there is no barrier BETWEEN steps — the workers start on batch i + 1 while the last dependent chains of batch i (deep
conflict trees, searches that run to the harness cap: ~95 ms during which a lone step leaves the device four-fifths
idle) finish.  ms_per_step = region / K.  `one_call_per_step` = the same steps again with one solver call each
(--sync-steps, after the timed region), and --no-stream times the region that way.

The timed region DELIVERS SCHEDULES: every solved instance's paths are written to the caller's buffers
(mrp_hl_solution.paths_xy, the equivalent of `solution = P.solution`, ecbs.hpp:238) inside it (config.delivers_schedules).

Extra objects in the same line:
  by_workload  — N = 1 only: one timed step each of the other shapes north_star names, after the headline region:
                 agents50, agents100 (synthetic, same generator; three / two batches as one stream), "shipped" (ALL 1000 shipped
                 benchmark/32x32_obst204 inputs, tests/golden/shipped_32x32.npz, as ONE batch, every result checked against
                 tests/golden/shipped_32x32_expected.json) and "shipped_heavy_tail" (agents100_ex36 with NO cap, against
                 tests/golden/shipped_heavy_tail_expected.json; --legs ...,shipped_heavy_tail: it runs for a minute or two and
                 is not part of the default).  Each carries value (expansions/s), instances_per_s, capped, cpu_baseline
                 (the CPU port, one thread, on a bounded sample of the SAME instances) and parity_mismatches_vs_gpu over
                 that sample.  sipp50 / sipp100 / sipp200: prioritized SIPP (config 5) on synthetic 64x64 maps with 410
                 obstacles, CPU leg = 512 instances timed inside the oracle.
  roofline     — bound = HBM (8 TB/s); achieved = 128 B/expansion x the job's expansions per second, i.e.
                 CHIP level: the resident launches of a step (one front + one heavy launch per host thread) run
                 concurrently, so a per-launch figure (kept in per_launch: bytes of a launch / its hipEvent duration)
                 says little.  traffic = HBM bytes per launch scaled from the committed PMC passes (bench.py cannot run
                 rocprofv3's counter passes on itself).  The fraction is tiny by construction — the path is a
                 latency- and issue-bound replay of sequential heaps whose working set sits in LDS / L2 —
                 so `pipes` rides along: per-pipe busy fractions of a LOADED chip from the committed PMC pass over the
                 batch-mode kernel (file named in pipes.source), not measured in this run.
  cpu_baseline — the oracle's CPU restatement (kind "port"; the reference needs Boost/yaml-cpp and cannot be built
                 here) timed single-threaded on a bounded sample of the headline workload on this box's host cores,
                 with a per-instance parity check against the GPU results; cpu_baseline_all_cores = the same port, one
                 instance per thread on every CPU this process may use, WALL-CLOCK, over the WHOLE first timed batch, every
                 instance compared with the GPU's result: status, cost, makespan, both expansion counters and the
                 64-bit digest of its schedule (parity_whole_first_batch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Every host worker thread keeps up to two resident kernels (front + heavy workgroups) on streams of their own, and each
# needs its own hardware queue (the ROCm default of 4 makes streams share queues and serialise their kernels — fatal for
# resident kernels, which only end when their host thread is done; measured on this box, scripts/micro/hw_queues.hip: 64
# streams hold 64 resident kernels at once with GPU_MAX_HW_QUEUES=64).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")

ALGO_BYTES_PER_EXPANSION = 128
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def usable_cpus():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a GPU
    box hands a job a share of the host, and threads beyond it only take turns)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 8
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def parity_count(per, digest, ra, hl):
    """Instances on which the CPU port (per [n][6] = rc, cost, makespan, hl, ll, ns; digest [n] uint64) and the GPU results
    (hl.BatchSolver.result_arrays) differ: solved ones in status, cost, makespan, both expansion counters and the
    schedule digest; capped on the CPU => capped on the GPU."""
    import numpy as np
    n = len(per)
    solved = per[:, 0] == 1
    ok = np.where(solved,
                  (ra["status"][:n] == hl.SOLVED) & (ra["cost"][:n] == per[:, 1]) & (ra["makespan"][:n] == per[:, 2]) &
                  (ra["hl_expanded"][:n] == per[:, 3]) & (ra["ll_expanded"][:n] == per[:, 4]) &
                  (ra["schedule_digest"][:n] == digest[:n]),
                  np.where(per[:, 0] == -1, ra["status"][:n] == hl.CAP, ra["status"][:n] == hl.NO_SOLUTION))
    return int((~ok).sum())


def cpu_leg(oracle, ia, ra, cap, n_sample, hl):
    """The CPU port, one thread, on the first n_sample instances of `ia`; per-instance parity against the GPU results."""
    n = min(n_sample, len(ia))
    per, digest, wall = oracle.mapf_solve_batch_digest(oracle.ECBS, ia.dimx, ia.dimy, ia.obstacles[:n], ia.starts[:n],
                                                       ia.goals[:n], w=1.3, cap_total=cap, n_threads=1)
    search_s = float(per[:, 5].sum()) / 1e9
    mism = parity_count(per, digest, ra, hl)
    # SURVEY.md §7 hard part 2: where Boost.Heap is installed the same sample also runs through the oracle built on the real
    # boost::heap::d_ary_heap ("identical" closes the w > 1 tie-break gap); this image has no Boost: "absent"
    boost = oracle.boost_crosscheck(oracle.ECBS, ia.dimx, ia.dimy, ia.obstacles[:n], ia.starts[:n], ia.goals[:n], per, w=1.3,
                                    cap_total=cap, n_threads=1)
    return {"value": float(per[:, 4].sum()) / max(search_s, 1e-12), "unit": "expansions/s", "cores": 1, "kind": "port",
            "boost_crosscheck": boost,
            "sample": "first %d instances of the leg's batch, oracle ECBS w=1.3 (g++ -O3), search() time only" % n,
            "instances_per_s": n / max(search_s, 1e-12), "seconds": search_s,
            "capped": int((per[:, 0] == -1).sum()), "parity_mismatches_vs_gpu": mism,
            "parity_fields": "status, cost, makespan, highLevelExpanded, lowLevelExpanded, schedule digest"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=262144, help="instances per GPU per step (headline workload)")
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--threads", type=int, default=0, help="host worker threads per GPU (0 = auto)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--lds-nodes", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=4096, help="headline instances timed on the CPU port (rank 0)")
    ap.add_argument("--path-cap", type=int, default=128, help="states per agent the caller's schedule buffers hold in the "
                                                               "headline region (a longer path would be reported, not hidden)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true",
                    help="one solver call per step with a barrier between steps (the default hands the K timed steps to the "
                         "solver as one stream of batches, mrp_hl_solver_solve_stream)")
    ap.add_argument("--sync-steps", type=int, default=2,
                    help="after the timed region: that many of its steps again, one call each (reported as one_call_per_step)")
    ap.add_argument("--legs", default="auto", help="'auto' (all at N=1, none otherwise), 'none', or a comma list of "
                                                   "agents50,agents100,shipped,sipp50,sipp100,sipp200")
    ap.add_argument("--max-ll-expansions", type=int, default=50000,
                    help="harness cap per instance of the headline workload (the reference has none and never returns "
                         "on infeasible inputs); applied identically to the GPU path and to the CPU baseline")
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

    from libmultirobotplanning_amd import hl, sharding
    hc = usable_cpus()
    # up to sixteen worker threads per GPU; the solver runs at most eight engines (each keeps two resident kernels — front
    # + heavy workgroups — and beyond ~20 hardware queues per process the device time-slices them), worker threads beyond
    # that share the engines two by two
    threads = args.threads or max(2, min(16, hc // max(1, world)))
    solver = hl.BatchSolver(device=local_rank, n_threads=threads, slots=args.slots, lds_nodes=args.lds_nodes)

    B, K, W = args.instances, args.steps, args.warmup

    def batch(step_idx):
        # distinct seeds per (rank, step); step indices of the timed steps start after the warm-up ones
        return hl.generate_instances(sharding.seed_base(args.agents, rank, K + W, step_idx, B), B, 32, 32, 204, args.agents)

    t_gen = time.perf_counter()
    batches = [batch(i) for i in range(K + W)]
    # marshal the batches and upload their static maps to HBM before the timed region (the reference constructs its
    # Environment before it starts its Timer, example/ecbs.cpp:576-582); the timed region is the searches only
    prepared = [solver.prepare(b, want_paths=True, path_cap=args.path_cap) for b in batches]
    t_gen = time.perf_counter() - t_gen

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
    rounds_total = 0
    stream = not args.no_stream and K > 1
    if stream:
        # the K steps as ONE stream of K batches: every batch is solved completely and its schedules delivered inside the
        # timed region, but the solver's workers go on to batch i + 1 while the last conflict trees of batch i finish
        st = solver.solve_stream(prepared[W:W + K], algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions)
        exp_total, solved_total, searches_total, rounds_total = (st["ll_expansions"], st["solved"], st["ll_searches"],
                                                                 st["rounds"])
    else:
        for i in range(W, W + K):
            _, st = solver.solve_prepared(prepared[i], algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions,
                                          raw=True)
            exp_total += st["ll_expansions"]
            solved_total += st["solved"]
            searches_total += st["ll_searches"]
            rounds_total += st["rounds"]
    barrier()
    elapsed = time.perf_counter() - t0
    lls = solver.ll_stats()
    # the same steps again, one solver call each (a barrier between steps): what the stream is worth
    sync_ref = None
    if stream and args.sync_steps > 0:
        ns = min(args.sync_steps, K)
        digest0 = solver.result_arrays(prepared[W])["schedule_digest"].copy()
        barrier()
        ts = time.perf_counter()
        es = 0
        for i in range(W, W + ns):
            _, st1 = solver.solve_prepared(prepared[i], algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions,
                                           raw=True)
            es += st1["ll_expansions"]
        barrier()
        ts = time.perf_counter() - ts
        import numpy as _np
        sync_ref = {"steps": ns, "value": es / max(ts, 1e-12), "ms_per_step": 1e3 * ts / ns,
                    "same_schedules_as_the_stream": bool(_np.array_equal(
                        digest0, solver.result_arrays(prepared[W])["schedule_digest"]))}
    first_ra = solver.result_arrays(prepared[W]) if K > 0 else None
    first_batch = batches[W] if K > 0 else None
    for i, p in enumerate(prepared):
        solver.release(p)
    capped_first = int((first_ra["status"] == hl.CAP).sum()) if first_ra is not None else 0
    longest_path = int(first_ra["path_len"].max()) if first_ra is not None else 0

    # totals: the only exchange this path needs (max of the elapsed times, sum of the counters)
    elapsed_max, (exp_all, solved_all, searches_all, inst_all) = sharding.reduce_totals(
        dist, "cpu" if rehearsal else "cuda", elapsed, [exp_total, solved_total, searches_total, K * B])

    # N > 1 only: ONE heavy conflict tree with the searches of every round sharded over all ranks: the
    # instance is broadcast from rank 0, each rank runs its share of a round on its own GPU, one all-gather per round
    sharded_ct = None
    if world > 1 and args.legs != "none":
        from libmultirobotplanning_amd import ct_sharded
        inst0 = None
        exp0 = None
        name0 = "map_32by32_obst204_agents100_ex0"
        if rank == 0:
            with open(os.path.join(ROOT, "tests", "golden", "bench_instances.json")) as f:
                inst0 = json.load(f)[name0]
            with open(os.path.join(ROOT, "tests", "golden", "oracle_expected.json")) as f:
                exp0 = json.load(f)[name0]["ecbs_w1.3"]
        dev = "cpu" if rehearsal else "cuda"
        run = None
        try:
            inst0 = ct_sharded.broadcast_instance(inst0, dist, dev)
            run = ct_sharded.gpu_executor(inst0, device=local_rank)
            legs_ct = {}
            for k in (1, world):  # look-ahead 1 = the plain loop (one node's two children per round), then one node per rank
                barrier()
                t1 = time.perf_counter()
                r = ct_sharded.solve_sharded(inst0, run, dist, algo=hl.ECBS, w=1.3, spec_width=k, device=dev)
                barrier()
                legs_ct["spec_width_%d" % k] = {"seconds": time.perf_counter() - t1, "rounds": r["rounds"],
                                                "searches_run_on_rank0": r["searches_run_here"]}
            if rank == 0:
                sharded_ct = {"workload": "ECBS w=1.3 on the shipped %s, every round's searches sharded over %d ranks" % (name0, world),
                              "cost": r["cost"], "hl_expanded": r["hl_expanded"], "ll_expanded": r["ll_expanded"],
                              "matches_golden": (r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
                                  exp0["cost"], exp0["makespan"], exp0["hl"], exp0["ll"]), **legs_ct}
        except Exception as e:  # the extra leg must never cost the contract line above it
            sharded_ct = {"error": repr(e)}
        finally:
            if run is not None:
                run.close()

    if rank == 0:
        kernel_s = lls["kernel_ms"] / 1e3
        per_launch_achieved = ALGO_BYTES_PER_EXPANSION * lls["expansions"] / max(kernel_s, 1e-12) / 1e9
        achieved = ALGO_BYTES_PER_EXPANSION * (exp_all / elapsed_max) / max(world, 1) / 1e9  # per GPU, all launches together
        traffic = None
        pipes = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")) as f:
                pmc = json.load(f)
            traffic = pmc["bytes_per_expansion_raw"] * lls["expansions"] / max(lls["launches"], 1)
            pipes = pmc.get("pipes")
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "low_level_node_expansions_per_sec",
            "value": exp_all / elapsed_max,
            "unit": "expansions/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed_max / max(K, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "ECBS w=1.3, synthetic 32x32_obst204-shaped instances, agents%d, %d instances/GPU/step "
                                   "(configs[1] shape)" % (args.agents, B),
                       "instances_per_gpu_per_step": B, "agents": args.agents, "host_threads_per_gpu": threads,
                       "host_cpus_usable": hc, "max_ll_expansions_per_instance": args.max_ll_expansions,
                       "delivers_schedules": True, "schedule_buffer_states_per_agent": args.path_cap,
                       "steps_as_one_stream": stream,
                       "longest_path_in_first_timed_step": longest_path,
                       "parallelism": "instances sharded per GPU, no data-path collective"},
            "instances_per_s": inst_all / elapsed_max,
            "solved": int(solved_all),
            "instances": int(inst_all),
            "capped_in_first_timed_step": capped_first,
            "ll_searches": int(searches_all),
            "setup_seconds_generate_and_preload": t_gen,
            "one_call_per_step": sync_ref,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "scaled from the committed PMC passes (profiles/hbm_traffic_pmc.json), raw counter bytes",
                "level": "chip: 128 B x the job's expansions per second per GPU (the step's resident launches overlap in time)",
                "kernel": "mrp_ll_ecbs_front_kernel (+ mrp_ll_ecbs_heavy_kernel)" if lls["heavy_active_wgs"] else "mrp_ll_ecbs_persistent_kernel",
                "per_launch": {"achieved": per_launch_achieved, "frac": per_launch_achieved / HBM_PEAK_GBS,
                               "launches": lls["launches"], "avg_launch_ms": lls["kernel_ms"] / max(lls["launches"], 1),
                               "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EXPANSION * lls["expansions"] / max(lls["launches"], 1),
                               "note": "rank-0 launches of the timed region (an ECBS session is a front and a heavy launch, both counted), "
                                       "hipEvent durations on their own streams; a launch stays resident for its engine's share of a step"},
                "pipes": pipes,
            },
            "tiers": {"front_us_per_expansion": lls["prof"][0] / 100.0 / max(lls["prof"][1], 1), "front_expansions": lls["prof"][1],
                      "beyond_front_us_per_expansion": lls["prof"][2] / 100.0 / max(lls["prof"][3], 1),
                      "beyond_front_expansions": lls["prof"][3], "searches_handed_over": lls["prof"][7],
                      "front_workgroups_busy_fraction": lls["session_busy_ms"] / max(lls["session_busy_ms"] + lls["session_idle_ms"], 1e-9),
                      "heavy_workgroups": lls["heavy_active_wgs"] // max(1 if stream else K, 1),
                      "heavy_workgroups_busy_fraction": lls["heavy_busy_ms"] / max(lls["heavy_busy_ms"] + lls["heavy_idle_ms"], 1e-9),
                      "tickets_per_step": int(rounds_total // max(K, 1))},
        }
        do_cpu = not args.no_cpu_baseline and world == 1  # the CPU legs run at N=1 only (rank 0)
        oracle = None
        if do_cpu:
            import oracle
            oracle.build()
            out["cpu_baseline"] = cpu_leg(oracle, first_batch, first_ra, args.max_ll_expansions, args.cpu_sample, hl)
            out["cpu_baseline"]["host_cpus"] = hc
            out["vs_cpu_port_1core"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-12)
            # SURVEY.md §8(d) also asks for "one instance per thread on all host cores": wall clock of a native pool on
            # every CPU this process may use, over enough instances to keep them busy for a second or so
            # ... over the WHOLE first timed batch: it is also the parity check of every schedule the step delivered
            n_all = len(first_batch)
            per, digest, wall = oracle.mapf_solve_batch_digest(oracle.ECBS, 32, 32, first_batch.obstacles[:n_all],
                                                               first_batch.starts[:n_all], first_batch.goals[:n_all], w=1.3,
                                                               cap_total=args.max_ll_expansions, n_threads=hc)
            out["cpu_baseline_all_cores"] = {
                "value": float(per[:, 4].sum()) / max(wall, 1e-12), "unit": "expansions/s", "cores": hc, "kind": "port",
                "nproc": hc, "pool_wall_seconds": wall, "instances_per_s": n_all / max(wall, 1e-12),
                "sample": "all %d instances of timed step 0, one instance per thread on %d threads (all CPUs this "
                          "process may use); expansions / pool wall-clock seconds" % (n_all, hc),
            }
            out["parity_whole_first_batch"] = {
                "instances": n_all, "mismatches": parity_count(per, digest, first_ra, hl),
                "fields": "status, cost, makespan, highLevelExpanded, lowLevelExpanded, 64-bit FNV-1a digest of the schedule"}
            out["vs_cpu_port_all_cores"] = out["value"] / max(out["cpu_baseline_all_cores"]["value"], 1e-12)

        legs = args.legs
        if legs == "auto":
            legs = "agents50,agents100,shipped,sipp50,sipp100,sipp200" if world == 1 else "none"
        by = {"agents%d" % args.agents: {"value": out["value"], "instances_per_s": out["instances_per_s"],
                                         "instances": int(inst_all), "capped": capped_first,
                                         "cap_per_instance": args.max_ll_expansions, "see": "top-level fields"}}
        leg_specs = {  # name -> (agents, instances per batch, cap per instance, CPU sample, batches in the stream)
            # batches large enough that the one-wavefront chains of the few pathological instances (the capped ones run
            # for seconds) do not leave the chip idle for most of the step; and, as in the headline region, a few batches
            # as one stream (scripts/stream_legs_probe.py: 3 x 65536 at fifty agents 6.2e8 against 4.95e8 one call each,
            # 2 x 16384 at a hundred 2.47e8 against 1.87e8)
            "agents50": (50, 65536, 400000, 192, 1 if args.no_stream else 3),
            "agents100": (100, 16384, 3000000, 24, 1 if args.no_stream else 2),
        }
        # (batch sizes: every instance is a chain of `agents` dependent searches, and the shortest chains need the largest
        # batch to keep the device's 4096 resident searches fed to the end — measured, scripts/sipp_bench.py: fifty agents
        # 6.0e8 / 6.8e8 / 7.5e8 at 8192 / 16384 / 32768 instances)
        sipp_specs = {"sipp50": (50, 32768, 512), "sipp100": (100, 8192, 512), "sipp200": (200, 8192, 512)}
        for name in [x for x in legs.split(",") if x and x != "none"]:
            if name in leg_specs:
                ag, nb, cap, ncpu, nbat = leg_specs[name]
                ia = hl.generate_instances(1000 * ag, nb, 32, 32, 204, ag)  # the first batch: the one the CPU legs sample
                more = [hl.generate_instances(1000 * ag + 100000 * b, nb, 32, 32, 204, ag) for b in range(1, nbat)]
                small = solver.prepare(ia[:64], want_paths=False)
                solver.solve_prepared(small, algo=hl.ECBS, w=1.3, max_ll_expansions=cap, raw=True)  # warm-up, same shape
                solver.release(small)
                preps = [solver.prepare(b, want_paths=False) for b in [ia] + more]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                st = solver.solve_stream(preps, algo=hl.ECBS, w=1.3, max_ll_expansions=cap)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                res = solver.result_arrays(preps[0])
                capped = sum(int((solver.result_arrays(p)["status"] == hl.CAP).sum()) for p in preps)
                for p in preps:
                    solver.release(p)
                del more
                leg = {"value": st["ll_expansions"] / dt, "unit": "expansions/s", "instances_per_s": nbat * nb / dt,
                       "instances": nbat * nb, "batches_in_the_stream": nbat, "seconds": dt, "solved": int(st["solved"]),
                       "capped": capped, "cap_per_instance": cap,
                       "ll_searches": int(st["ll_searches"]),
                       "workload": "ECBS w=1.3, synthetic 32x32_obst204-shaped, agents%d, %d batches of %d as one stream, "
                                   "seeds %d.. (+100000 per batch)" % (ag, nbat, nb, 1000 * ag)}
                if do_cpu:
                    leg["cpu_baseline"] = cpu_leg(oracle, ia, res, cap, ncpu, hl)
                    leg["vs_cpu_port_1core"] = leg["value"] / max(leg["cpu_baseline"]["value"], 1e-12)
                    n_all = min(nb, 8 * hc if ag <= 50 else 4 * hc)  # a bounded sample: ~10 s of wall clock on all cores
                    per, wall = oracle.mapf_solve_batch(oracle.ECBS, 32, 32, ia.obstacles[:n_all], ia.starts[:n_all],
                                                        ia.goals[:n_all], w=1.3, cap_total=cap, n_threads=hc)
                    leg["cpu_baseline_all_cores"] = {
                        "value": float(per[:, 4].sum()) / max(wall, 1e-12), "unit": "expansions/s", "cores": hc,
                        "kind": "port", "pool_wall_seconds": wall,
                        "sample": "first %d instances of the leg's batch, one per thread on %d threads" % (n_all, hc)}
                    leg["vs_cpu_port_all_cores"] = leg["value"] / max(leg["cpu_baseline_all_cores"]["value"], 1e-12)
                by[name] = leg
            elif name in sipp_specs:
                # BASELINE.json config 5 / SURVEY.md §8(d)(iv): prioritized planning over MRP_LL_SIPP searches, 64x64 with
                # 410 obstacles; every instance plans its agents one after the other (mapf_prioritized_sipp.cpp:214-270),
                # so a step is nb instances x ag sequential searches, the safe-interval tables resident on the device
                ag, nb, ncpu = sipp_specs[name]
                ia = hl.generate_instances(640000 + 1000 * ag, nb, 64, 64, 410, ag)
                insts = list(ia)
                solver.prioritized_sipp(insts, want_schedules=False)  # warm-up pass over the same batch (device table pool grows)
                torch.cuda.synchronize()
                res, st = solver.prioritized_sipp(insts, want_schedules=False)
                torch.cuda.synchronize()
                # the native driver's own clock around the whole batch (first submit to last result); the Python wrapper
                # around it only marshals the instance arrays in and the per-agent schedules out
                dt = st["wall_seconds"]
                leg = {"value": st["ll_expansions"] / dt, "unit": "expansions/s", "instances_per_s": nb / dt,
                       "instances": nb, "seconds": dt, "ll_searches": int(st["ll_searches"]),
                       "all_agents_planned": int(st["solved"]),
                       "workload": "prioritized SIPP, synthetic 64x64 with 410 obstacles, agents%d, seeds %d.." % (
                           ag, 640000 + 1000 * ag)}
                if do_cpu:
                    nt = max(1, min(hc, 8))
                    per, wall = oracle.prioritized_sipp_batch(64, 64, ia.obstacles[:ncpu], ia.starts[:ncpu], ia.goals[:ncpu],
                                                              n_threads=nt)
                    secs = float(per[:, 3].sum()) / 1e9
                    mism = sum((int(q[1]), int(q[0]), int(q[2])) != (r["cost"], r["n_planned"], r["expanded"])
                               for q, r in zip(per, res[:ncpu]))
                    leg["cpu_baseline"] = {
                        "value": float(per[:, 2].sum()) / max(secs, 1e-12), "unit": "expansions/s", "cores": 1, "kind": "port",
                        "sample": "first %d instances of the leg's batch; each instance timed inside the oracle (g++ -O3), "
                                  "%d at a time; expansions / sum of the per-instance seconds" % (ncpu, nt),
                        "instances_per_s": ncpu / max(secs, 1e-12), "seconds": secs, "pool_wall_seconds": wall,
                        "parity_mismatches_vs_gpu": int(mism)}
                    leg["vs_cpu_port_1core"] = leg["value"] / max(leg["cpu_baseline"]["value"], 1e-12)
                by[name] = leg
            elif name == "shipped_heavy_tail":
                # the one shipped input the golden file's cap cuts off, run to completion (the reference has no cap): one deep
                # conflict tree, i.e. a chain of ~70 000 dependent rounds of two searches each — the GPU adds nothing to a
                # single chain but must produce the same answer
                gold = os.path.join(ROOT, "tests", "golden", "shipped_heavy_tail_expected.json")
                with open(gold) as f:
                    exp36 = json.load(f)["map_32by32_obst204_agents100_ex36"]
                corpus = dict(hl.load_shipped_corpus(os.path.join(ROOT, "tests", "golden", "shipped_32x32.npz")))
                inst = corpus["map_32by32_obst204_agents100_ex36"]
                prep = solver.prepare([inst], want_paths=True, path_cap=1024)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, st = solver.solve_prepared(prep, algo=hl.ECBS, w=1.3, max_ll_expansions=-1, raw=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                r = solver.results_of(prep)[0]
                solver.release(prep)
                import hashlib
                hsh = hashlib.sha256()
                for pth in r.get("paths", []):
                    hsh.update(("|" + ",".join("%d:%d" % (x, y) for x, y in pth)).encode())
                got = dict(rc=1 if r["status"] == hl.SOLVED else 0, cost=r["cost"], makespan=r["makespan"], hl=r["hl_expanded"],
                           ll=r["ll_expanded"], digest=hsh.hexdigest()[:16])
                by[name] = {"value": st["ll_expansions"] / dt, "unit": "expansions/s", "seconds": dt, "instances": 1,
                            "result": got, "matches_golden": all(got[k] == exp36[k] for k in got),
                            "cpu_baseline": {"kind": "port", "cores": 1, "seconds": exp36.get("oracle_search_seconds_in_the_build_container"),
                                             "sample": "the same instance, our oracle uncapped, timed when the golden vector "
                                                       "was generated (tests/golden/make_fixtures.py --heavy-tail) in the build "
                                                       "container, not on this box"},
                            "workload": "shipped map_32by32_obst204_agents100_ex36, ECBS w=1.3, no cap"}
            elif name == "shipped":
                # the corpus north_star names: ALL 1000 shipped benchmark/32x32_obst204 inputs (agents10..100 x ex0..99) as
                # ONE batch, every result checked against tests/golden/shipped_32x32_expected.json (our oracle at the same
                # cap; agents100_ex36 runs into it on both sides)
                corpus = hl.load_shipped_corpus(os.path.join(ROOT, "tests", "golden", "shipped_32x32.npz"))
                with open(os.path.join(ROOT, "tests", "golden", "shipped_32x32_expected.json")) as f:
                    expected = json.load(f)
                names = [n for n, _ in corpus]
                insts = [i for _, i in corpus]
                cap = 3000000  # the cap the golden vectors were produced with (tests/golden/make_fixtures.py)
                prep = solver.prepare(insts, want_paths=False)
                solver.solve_prepared(prep, algo=hl.ECBS, w=1.3, max_ll_expansions=cap, raw=True)  # warm-up
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, st = solver.solve_prepared(prep, algo=hl.ECBS, w=1.3, max_ll_expansions=cap, raw=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                res = solver.results_of(prep)
                solver.release(prep)
                mism = 0
                for n, r in zip(names, res):
                    e = expected[n]
                    if e["rc"] == 1:
                        mism += (r["status"], r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) != (
                            hl.SOLVED, e["cost"], e["makespan"], e["hl"], e["ll"])
                    else:
                        mism += r["status"] != hl.CAP
                leg = {"value": st["ll_expansions"] / dt, "unit": "expansions/s", "instances_per_s": len(insts) / dt,
                       "instances": len(insts), "seconds": dt, "solved": int(st["solved"]),
                       "capped": sum(1 for r in res if r["status"] == hl.CAP), "cap_per_instance": cap,
                       "ll_searches": int(st["ll_searches"]),
                       "parity_mismatches_vs_golden": int(mism), "parity_checked": len(insts),
                       "workload": "all %d shipped benchmark/32x32_obst204 inputs (agents10..100 x ex0..99) as one batch, "
                                   "ECBS w=1.3" % len(insts)}
                if do_cpu:
                    # one core: every tenth input of every agent count (100 instances, ~15 s of CPU; agents100_ex36, which
                    # alone costs the port ~20 s to reach the cap, is not among them), each checked against the golden vector
                    t_cpu = 0.0
                    e_cpu = 0
                    n_cpu = 0
                    cpu_mism = 0
                    for n, inst in corpus:
                        if int(n.rsplit("_ex", 1)[1]) % 10 != 0:
                            continue
                        o = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=cap, path_cap=1024)
                        t_cpu += o["elapsed_ns"] / 1e9
                        e_cpu += o["ll_expanded"]
                        n_cpu += 1
                        e = expected[n]
                        cpu_mism += (o["rc"], o.get("cost"), o["ll_expanded"]) != (1, e.get("cost"), e.get("ll"))
                    leg["cpu_baseline"] = {"value": e_cpu / max(t_cpu, 1e-12), "unit": "expansions/s", "cores": 1,
                                           "kind": "port", "sample": "ex0, ex10, .. ex90 of every agent count (%d instances)" % n_cpu,
                                           "instances_per_s": n_cpu / max(t_cpu, 1e-12), "seconds": t_cpu,
                                           "golden_mismatches": int(cpu_mism)}
                    leg["vs_cpu_port_1core"] = leg["value"] / max(leg["cpu_baseline"]["value"], 1e-12)
                    # every core: the whole corpus, one instance per thread, heaviest agent counts first (wall clock)
                    wall_all = 0.0
                    e_all = 0
                    for nag in range(100, 9, -10):
                        grp = [i for n, i in corpus if ("agents%d_" % nag) in n]
                        ob = [i["obstacles"] for i in grp]
                        per, wall = oracle.mapf_solve_batch(oracle.ECBS, 32, 32, ob, [i["starts"] for i in grp],
                                                            [i["goals"] for i in grp], w=1.3, cap_total=cap, n_threads=hc)
                        wall_all += wall
                        e_all += int(per[:, 4].sum())
                    leg["cpu_baseline_all_cores"] = {"value": e_all / max(wall_all, 1e-12), "unit": "expansions/s",
                                                     "cores": hc, "kind": "port", "pool_wall_seconds": wall_all,
                                                     "sample": "all %d instances, one per thread on %d threads, agent "
                                                               "counts one after the other" % (len(insts), hc)}
                    leg["vs_cpu_port_all_cores"] = leg["value"] / max(leg["cpu_baseline_all_cores"]["value"], 1e-12)
                by[name] = leg
        if sharded_ct is not None:
            by["sharded_conflict_tree"] = sharded_ct
        out["by_workload"] = by
        # How much of the GPU a given number of host threads can feed (an 8-GPU job on this pool's 16-CPU quota has two
        # per GPU): one short step of the headline shape per thread count, each with its own solver (one engine, one
        # resident launch per thread; the chip's resident wavefronts are divided among the engines)
        if world == 1 and args.legs != "none" and first_batch is not None:
            sweep = {}
            solver.close()
            solver = None
            nb = min(B, 65536)
            sub = first_batch[:nb]
            for t in (16, 8, 4, 2):
                if t > max(hc, 2):
                    continue
                sv = hl.BatchSolver(device=local_rank, n_threads=t, slots=min(2048, max(512, 2048 // t)))
                try:
                    prep = sv.prepare(sub, want_paths=False)
                    sv.solve_prepared(prep, algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions, raw=True)
                    times = []
                    for _ in range(2):  # two timed steps, the faster one counts (both are listed)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        _, st = sv.solve_prepared(prep, algo=hl.ECBS, w=1.3, max_ll_expansions=args.max_ll_expansions, raw=True)
                        torch.cuda.synchronize()
                        times.append(time.perf_counter() - t1)
                    dt = min(times)
                    sv.release(prep)
                    sweep[str(t)] = {"value": st["ll_expansions"] / dt, "seconds": dt, "seconds_each": times, "instances": nb}
                finally:
                    sv.close()
            best = max(v["value"] for v in sweep.values()) if sweep else 0.0
            for t in sweep:  # (sixteen threads = sixteen engines of which eight run: an ECBS worker keeps two resident kernels)
                sweep[t]["vs_best"] = sweep[t]["value"] / max(best, 1e-12)
            out["host_threads_sweep"] = sweep
        print(json.dumps(out), flush=True)
    if solver is not None:
        solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
