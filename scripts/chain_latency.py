"""Latency of one conflict-tree round (host <-> resident kernel) on dependent chains: synthetic agents10 instances that run
into the expansion cap through thousands of tiny searches, solved alone (dev tool, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import hl
ia = hl.generate_instances(10000, 20000, 32, 32, 204, 10)
s = hl.BatchSolver(device=0, n_threads=int(sys.argv[1]) if len(sys.argv) > 1 else 2, slots=512)
for k in (5169, 9479, 3493):
    insts = [ia[k]]
    s.solve(insts, algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=50000)
    res, st = s.solve(insts, algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=50000)
    print("instance %d alone: wall %.1f ms, searches %d, rounds %d, expansions %d -> %.1f us per round, %.2f us per expansion" % (
        k, st["wall_seconds"] * 1e3, st["ll_searches"], st["rounds"], st["ll_expansions"],
        st["wall_seconds"] * 1e6 / max(st["rounds"], 1), st["wall_seconds"] * 1e6 / max(st["ll_expansions"], 1)), flush=True)
s.close()
