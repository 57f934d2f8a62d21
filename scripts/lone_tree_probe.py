"""One deep conflict tree alone on the GPU: wall time per conflict-tree round (the dependent chain that ends a step).
usage: lone_tree_probe.py [instance index in the agents10 bench batch] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import hl
k = int(sys.argv[1]) if len(sys.argv) > 1 else 9479
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
insts = hl.generate_instances(10000, k + 1, 32, 32, 204, 10)
one = [insts[k]]
s = hl.BatchSolver(device=0, n_threads=threads, slots=int(os.environ.get("MRP_SLOTS", "512")))
for rep in range(int(os.environ.get("MRP_REPS", "3"))):
    s.ll_stats(reset=True)
    res, st = s.solve(one, algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=50000)
    r = res[0]
    print("rep %d: wall %.1f ms status %d HL %d LL %d searches %d rounds %d -> %.1f us per HL expansion, %.1f us per round" % (
        rep, st["wall_seconds"] * 1e3, r["status"], r["hl_expanded"], r["ll_expanded"], st["ll_searches"], st["rounds"],
        st["wall_seconds"] * 1e6 / max(r["hl_expanded"], 1), st["wall_seconds"] * 1e6 / max(st["rounds"], 1)), flush=True)
    ls = s.ll_stats()
    pf = ls["prof"]
    print("   device: %.2f us/expansion narrow over %d, %.1f us per job over %d jobs" % (
        pf[0] / 100.0 / max(pf[1], 1), pf[1], pf[4] / 100.0 / max(pf[5], 1), pf[5]), flush=True)
