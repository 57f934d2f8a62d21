import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll
d = json.load(open(os.path.join(ROOT, "tests/golden/bench_instances.json")))
inst = d["map_32by32_obst204_agents10_ex0"]
_, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3)
for lds_nodes, arena in ((32, 0), (32, 65536), (600, 65536), (0, 65536), (-1, 65536)):
    eng = ll.LowLevelEngine(device=0, lds_nodes=lds_nodes, arena_nodes=arena, n_tickets=1, slots=4)
    mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
    jobs = [ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                     agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                     edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]) for c in calls]
    res = eng.search_batch(jobs)
    print("lds_nodes", lds_nodes, "arena", arena)
    for c, r in zip(calls, res):
        print("   want exp %d cost %d | got status %d exp %d cost %d tier %d nstates %d" % (c["expanded"], c["cost"], r.status, r.expanded, r.cost, r.tier, len(r.states)))
    print("   stats", {k: v for k, v in eng.stats().items() if k in ("jobs", "expansions", "migrated", "prof")})
    eng.close()
