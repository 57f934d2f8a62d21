"""Throughput probe of the prioritized-SIPP driver (BASELINE.json config 5 shape: 64x64, 10 % obstacles)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
import oracle
from libmultirobotplanning_amd import hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cpu_n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
s = hl.BatchSolver(device=0, n_threads=threads, slots=512)
ia = hl.generate_instances(640000 + 1000 * agents, n, 64, 64, 410, agents)
insts = list(ia)
s.prioritized_sipp(insts[:64])
for rep in range(2):
    s.ll_stats(reset=True)
    res, st = s.prioritized_sipp(insts)
    ls = s.ll_stats()
    print("   engine totals over %d threads: launches %d, kernel %.1f ms, pack %.1f ms, unpack %.1f ms; resident workgroups busy %.1f s, waiting %.1f s" % (
        threads, ls["launches"], ls["kernel_ms"], ls["pack_ms"], ls["unpack_ms"], ls["session_busy_ms"] / 1e3,
        ls["session_idle_ms"] / 1e3), flush=True)
    pf = ls["prof"]
    print("   SIPP kernel: LDS tier %.2f us/expansion over %.3g expansions, arena tier %.2f us/expansion over %.3g; runSipp %.1f us/job over %d jobs" % (
        pf[0] / 100.0 / max(pf[1], 1), pf[1], pf[2] / 100.0 / max(pf[3], 1), pf[3], pf[4] / 100.0 / max(pf[5], 1), pf[5]), flush=True)
    print("   middle tier (open list in LDS, nodes in the arena): %.2f us/expansion over %.3g expansions" % (pf[6] / 100.0 / max(pf[7], 1), pf[7]))
    print("rep %d: %d instances x %d agents: wall %.3f s, %.3e exp/s, %.1f inst/s, rounds %d searches %d planned-all %d" % (
        rep, n, agents, st["wall_seconds"], st["ll_expansions"] / st["wall_seconds"], n / st["wall_seconds"], st["rounds"],
        st["ll_searches"], st["solved"]), flush=True)
if os.environ.get("MRP_NO_CPU"):
    sys.exit(0)
cpu_n = min(max(cpu_n, 512), n)
per, wall = oracle.prioritized_sipp_batch(64, 64, ia.obstacles[:cpu_n], ia.starts[:cpu_n], ia.goals[:cpu_n], n_threads=1)
mism = sum((int(p[1]), int(p[0]), int(p[2])) != (r["cost"], r["n_planned"], r["expanded"]) for p, r in zip(per, res[:cpu_n]))
secs = per[:, 3].sum() / 1e9
print("cpu port (1 thread, timed inside the oracle) on the first %d instances: %.3e exp/s, %.2f inst/s, mismatches %d" % (
    cpu_n, per[:, 2].sum() / secs, cpu_n / secs, mism))
