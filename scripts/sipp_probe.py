"""Per-expansion cost of the SIPP kernel: NT identical device-resident tables, one job each, through a session of WG
workgroups.  usage: sipp_probe.py [workgroups] [tables] [special_cells]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import ll
wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nspec = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
rng = random.Random(5)
dim = 64
obst = [list(c) for c in {(rng.randrange(dim), rng.randrange(dim)) for _ in range(410)}]
oset = {tuple(c) for c in obst}
free = [[x, y] for x in range(dim) for y in range(dim) if (x, y) not in oset]
eng = ll.LowLevelEngine(device=0, max_cells=dim * dim, slots=2048)
mid = eng.upload_map(dim, dim, obst)
cells = rng.sample(free, nspec)
ivs = [(c, rng.randrange(0, 120)) for c in cells]
tabs = []
for k in range(nt):
    h = eng.sipp_table_create(mid)
    for c, t in ivs:
        eng.sipp_table_add(h, c[0], c[1], t, t + 1)
    tabs.append(h)
pairs = [(rng.choice(free), rng.choice(free)) for _ in range(nt)]
eng.session_begin_sipp(wgs)
try:
    for rep in range(3):
        jobs = [ll.LLJob(map_id=mid, algo=ll.SIPP, start=s, goal=g, sipp_table=h) for (s, g), h in zip(pairs, tabs)]
        t0 = time.perf_counter()
        res = eng.search_batch(jobs)
        dt = time.perf_counter() - t0
        ex = sum(r.expanded for r in res)
        tiers = sorted({r.tier for r in res})
        print("wgs %d rep %d: %d jobs, %d expansions, %.3f ms wall -> %.2f us / expansion / workgroup, %.3e exp/s; tiers %s" % (
            wgs, rep, len(jobs), ex, dt * 1e3, dt * 1e6 * min(wgs, nt) / ex, ex / dt, tiers), flush=True)
finally:
    eng.session_end()
st = eng.stats()
print("   busy %.1f ms, idle %.1f ms -> %.2f us busy per expansion" % (st["session_busy_ms"], st["session_idle_ms"], st["session_busy_ms"] * 1e3 / (3 * ex)))
print("total_expansions %d" % (3 * ex))
