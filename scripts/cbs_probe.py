"""CBS 8x8 probe (dev tool): python scripts/cbs_probe.py [agents] [instances]; use with MRP_HL_TIMING=1 / MRP_HL_DEEP=<n>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = hl.BatchSolver(device=0, n_threads=16, slots=512)
insts = [hl.generate_instance(800000 + 1000 * agents + k, 8, 8, 12, agents) for k in range(n)]
s.solve(insts[:256], algo=hl.CBS, max_ll_expansions=100000, want_paths=False)
for rep in range(2):
    s.ll_stats(reset=True)
    res, st = s.solve(insts, algo=hl.CBS, max_ll_expansions=100000, want_paths=False)
    ls = s.ll_stats()
    print("rep %d: wall %.3f s, %.3e exp/s, searches %d, rounds %d; workgroups busy %.1f s waiting %.1f s" % (
        rep, st["wall_seconds"], st["ll_expansions"] / st["wall_seconds"], st["ll_searches"], st["rounds"],
        ls["session_busy_ms"] / 1e3, ls["session_idle_ms"] / 1e3), flush=True)
