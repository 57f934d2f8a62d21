"""Latency of one hard instance's conflict-tree chain (rounds are sequential): wall / rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from libmultirobotplanning_amd import hl
s = hl.BatchSolver(device=0, n_threads=1, slots=64)
for seed in (10000 + 5169, 10000 + 9479, 10000 + 3493):
    inst = hl.generate_instance(seed, 32, 32, 204, 10)
    for mode in (0, 1):
        s.ll_stats(reset=True)
        res, st = s.solve([inst], algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=50000, mode=mode)
        ls = s.ll_stats()
        r = res[0]
        print("seed %d mode %d: wall %.3f s, HL %d, LL searches %d, LL exp %d -> %.1f us/HL-round, %.2f us/expansion ; busy %.3f s" % (
            seed, mode, st["wall_seconds"], r["hl_expanded"], r["ll_searches"], r["ll_expanded"],
            1e6 * st["wall_seconds"] / max(r["hl_expanded"], 1), 1e6 * st["wall_seconds"] / max(r["ll_expanded"], 1),
            ls["session_busy_ms"] / 1e3), flush=True)
