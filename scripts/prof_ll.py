"""Cycle breakdown of the low-level kernel (needs the -DMRP_LL_TRACE library: MRP_LL_LIB=.../libmrp_ll_trace.so)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll, hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lds_nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=1024, lds_nodes=lds_nodes)
jobs, exp = [], []
for k in range(n_inst):
    inst = hl.generate_instance(1000 * agents + k, 32, 32, 204, agents)
    summary, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3)
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
    for c in calls:
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
    print("rep %d wall %.1f ms kernel %.2f ms; migrated %d; cycles/expansion %.0f" % (rep, dt * 1e3, st["kernel_ms"], st["migrated"], p[5] / sum(exp)))
    names = ["walk", "pop+erase", "push", "entries", "top(incl walk)", "total", "#walks", "walk-visited"]
    print("  " + "  ".join("%s=%.1f%%" % (names[i], 100.0 * p[i] / tot) for i in range(5)))
    print("  walks %d, visited %d (%.1f per walk), cycles per visited %.0f; per-expansion: pop+erase %.0f push %.0f succ %.0f" % (
        p[6], p[7], p[7] / max(p[6], 1), p[0] / max(p[7], 1), p[1] / sum(exp), p[2] / sum(exp), p[3] / sum(exp)))
