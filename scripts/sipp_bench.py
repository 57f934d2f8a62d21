"""Throughput probe of the prioritized-SIPP driver (BASELINE.json config 5 shape: 64x64, 10 % obstacles)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import oracle
from libmultirobotplanning_amd import hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cpu_n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
s = hl.BatchSolver(device=0, n_threads=threads, slots=512)
insts = [hl.generate_instance(640000 + 1000 * agents + k, 64, 64, 410, agents) for k in range(n)]
s.prioritized_sipp(insts[:64])
for rep in range(2):
    s.ll_stats(reset=True)
    res, st = s.prioritized_sipp(insts)
    ls = s.ll_stats()
    print("   engine totals over %d threads: launches %d, kernel %.1f ms, pack %.1f ms, unpack %.1f ms; resident workgroups busy %.1f s, waiting %.1f s" % (
        threads, ls["launches"], ls["kernel_ms"], ls["pack_ms"], ls["unpack_ms"], ls["session_busy_ms"] / 1e3,
        ls["session_idle_ms"] / 1e3), flush=True)
    print("rep %d: %d instances x %d agents: wall %.3f s, %.3e exp/s, %.1f inst/s, rounds %d searches %d planned-all %d" % (
        rep, n, agents, st["wall_seconds"], st["ll_expansions"] / st["wall_seconds"], n / st["wall_seconds"], st["rounds"],
        st["ll_searches"], st["solved"]), flush=True)
t = 0.0; e = 0; mism = 0
for inst, r in zip(insts[:cpu_n], res[:cpu_n]):
    t0 = time.perf_counter()
    o = oracle.prioritized_sipp(inst)
    t += time.perf_counter() - t0
    e += o["expanded"]
    mism += (o["cost"], o["planned"], o["expanded"]) != (r["cost"], r["planned"], r["expanded"])
print("cpu oracle on first %d: %.3e exp/s, %.2f inst/s (incl. python wrapper), mismatches %d" % (cpu_n, e / t, cpu_n / t, mism))
