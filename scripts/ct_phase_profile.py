"""Shader cycles per phase of the compact tier's search loop (dev tool, GPU box).  Needs the diagnostic library:
   python -c "from libmultirobotplanning_amd import _build; _build.build_variant('ctprof', ['-DMRP_CT_PROF'])"
   MRP_LL_LIB=libmultirobotplanning_amd/lib/libmrp_ll_ctprof.so python scripts/ct_phase_profile.py [agents] [instances] [slots]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll, hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 64
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=slots)
jobs, exp = [], []
for k in range(n_inst):
    inst = hl.generate_instance(1000 * agents + k, 32, 32, 204, agents)
    _, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3)
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
    for c in calls:
        jobs.append(ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                             agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]))
        exp.append(c["expanded"])
eng.search_batch(jobs[:32])
eng.reset_stats()
res = eng.search_batch(jobs)
assert [r.expanded for r in res] == exp
p = eng.stats()["prof"]
E = float(sum(exp))
names = ["loop top + goal test", "ordered walk", "probes / row loads issued", "pop + erase", "successor entries", "pushes",
         "(walk-visited nodes)", "set-up"]
tot = sum(p[k] for k in (0, 1, 2, 3, 4, 5, 7))
print("jobs %d expansions %d; shader cycles per expansion %.0f (with the probes' own cost)" % (len(jobs), E, tot / E))
for k in (0, 1, 2, 3, 4, 5, 7):
    print("  %-28s %8.1f cycles/expansion  %5.1f %%" % (names[k], p[k] / E, 100.0 * p[k] / tot))
print("  walk-visited nodes per expansion %.2f; cycles per visited node %.0f" % (p[6] / E, p[1] / max(p[6], 1)))
eng.close()
