"""Does a stream of batches (mrp_hl_solver_solve_stream) pay at fifty / a hundred agents?  usage: agents instances cap n_batches"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
from libmultirobotplanning_amd import hl
ag, nb, cap, k = (int(x) for x in sys.argv[1:5])
s = hl.BatchSolver(device=0, n_threads=16)
batches = [hl.generate_instances(1000 * ag + 100000 * b, nb, 32, 32, 204, ag) for b in range(k)]
small = s.prepare(batches[0][:64], want_paths=False)
s.solve_prepared(small, algo=hl.ECBS, w=1.3, max_ll_expansions=cap, raw=True)
s.release(small)
preps = [s.prepare(b, want_paths=False) for b in batches]
t = time.perf_counter()
st = s.solve_stream(preps, algo=hl.ECBS, w=1.3, max_ll_expansions=cap)
dt = time.perf_counter() - t
print("stream of %d x %d agents%d: %.3f s, %.4g expansions/s" % (k, nb, ag, dt, st["ll_expansions"] / dt), flush=True)
e = 0
t = time.perf_counter()
for p in preps:
    _, s1 = s.solve_prepared(p, algo=hl.ECBS, w=1.3, max_ll_expansions=cap, raw=True)
    e += s1["ll_expansions"]
dt = time.perf_counter() - t
print("one call per batch: %.3f s, %.4g expansions/s (same expansions: %s)" % (dt, e / dt, e == st["ll_expansions"]), flush=True)
