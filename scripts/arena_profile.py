"""Cycle breakdown of the ARENA tier on the searches that outgrow the compact tier (dev tool, GPU box; needs the
-DMRP_LL_TRACE library: MRP_LL_LIB=.../libmrp_ll_trace.so).  usage: arena_profile.py [agents] [instances] [min expansions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll, hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
min_exp = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=256)
jobs, exp = [], []
for k in range(n_inst):
    inst = hl.generate_instance(1000 * agents + k, 32, 32, 204, agents)
    _, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3, cap_total=50000)
    big = [c for c in calls if c["expanded"] >= min_exp]
    if not big:
        continue
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
    for c in big:
        jobs.append(ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                             agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]))
        exp.append(c["expanded"])
print("jobs", len(jobs), "expansions", sum(exp), "max", max(exp), flush=True)
for rep in range(2):
    eng.reset_stats()
    t0 = time.time()
    res = eng.search_batch(jobs)
    dt = time.time() - t0
    st = eng.stats()
    assert [r.expanded for r in res] == exp
    p = st["prof"]
    tot = max(p[5], 1)
    E = float(sum(exp))
    print("rep %d wall %.1f ms; in the arena tier: %d of %d; cycles per expansion %.0f" % (rep, dt * 1e3, st["migrated"], len(jobs), p[5] / E))
    names = ["walk", "pop+erase", "push", "entries", "top(incl walk)"]
    print("  " + "  ".join("%s=%.1f%%" % (names[i], 100.0 * p[i] / tot) for i in range(5)))
    print("  walks %d, visited %d (%.1f per walk, %.2f per expansion), cycles per visited %.0f; per expansion: pop+erase %.0f push %.0f succ %.0f" % (
        p[6], p[7], p[7] / max(p[6], 1), p[7] / E, p[0] / max(p[7], 1), p[1] / E, p[2] / E, p[3] / E))
