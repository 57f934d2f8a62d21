"""Staged GPU diagnostic of the low-level engine; prints progress (flushed) after each step."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def say(*a):
    print("[%.2f]" % time.time(), *a, flush=True)
say("start")
from libmultirobotplanning_amd import ll
say("imported")
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
say("engine created")
mid = eng.upload_map(4, 3, [])
say("map uploaded")
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR, start=[0, 0], goal=[0, 0], max_expansions=1000)])
say("start==goal A*:", r)
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR, start=[0, 0], goal=[3, 2], max_expansions=1000)])
say("A* simple:", r)
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, w=1.3, start=[0, 0], goal=[3, 2], max_expansions=1000)])
say("EPS simple:", r)
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, w=1.3, start=[0, 0], goal=[3, 2], max_expansions=1000,
                               ctx_paths=[[], [[1, 0], [0, 0], [0, 1]], [[3, 2]]])])
say("EPS ctx:", r)
import oracle
inst = json.load(open(os.path.join(ROOT, "tests/golden/bench_instances.json")))["map_32by32_obst204_agents10_ex0"]
summary, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3)
mid2 = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
for c in calls:
    j = ll.LLJob(map_id=mid2, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                 agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                 edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"], max_expansions=200000)
    t0 = time.time()
    r = eng.search_batch([j])[0]
    say("agent", c["agent"], "status", r.status, "exp", r.expanded, "oracle", c["expanded"], "cost", r.cost, c["cost"],
        "tier", r.tier, "ms %.2f" % ((time.time() - t0) * 1e3), "path_ok", [s[1:] for s in r.states] == c["states"])
say(eng.stats())
