"""Harvested ECBS low-level searches (agents10 shape) as ONE batch-mode launch that keeps `slots` searches resident: the
workload of the PMC passes over a LOADED chip (scripts/r4_pmc_loaded.sh).  usage: pmc_jobs.py [instances] [slots]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll, hl
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=slots)
jobs, exp = [], []
for k in range(n_inst):
    inst = hl.generate_instance(10000 + k, 32, 32, 204, 10)
    _, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3, cap_total=50000)
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
    for c in calls:
        jobs.append(ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                             agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                             edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]))
        exp.append(c["expanded"])
jobs, exp = jobs * 4, exp * 4  # every search four times: a launch long enough for the chip to stay full
print("jobs", len(jobs), "expansions", sum(exp), "max", max(exp), "slots", slots, flush=True)
for rep in range(2):
    eng.reset_stats()
    t0 = time.time()
    res = eng.search_batch(jobs)
    dt = time.time() - t0
    st = eng.stats()
    assert [r.expanded for r in res] == exp
    p = st["prof"]
    print("rep %d wall %.1f ms kernel %.2f ms; LDS tier %.3f us/expansion over %d; arena %.3f us/expansion over %d" % (
        rep, dt * 1e3, st["kernel_ms"], p[0] / 100.0 / max(p[1], 1), p[1], p[2] / 100.0 / max(p[3], 1), p[3]), flush=True)
