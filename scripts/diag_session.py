"""Staged diagnostic of session mode (resident kernel + host job ring)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def say(*a):
    print("[%.3f]" % time.time(), *a, flush=True)
from libmultirobotplanning_amd import ll
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
mid = eng.upload_map(4, 3, [])
say("batch mode:", eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR, start=[0, 0], goal=[3, 2], max_expansions=1000)])[0].cost)
eng.session_begin(8)
say("session begun")
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR, start=[0, 0], goal=[3, 2], max_expansions=1000)])
say("session job:", r[0])
r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, w=1.3, start=[0, 0], goal=[3, 2], max_expansions=1000,
                               ctx_paths=[[], [[1, 0], [0, 0], [0, 1]], [[3, 2]]]) for _ in range(20)])
say("session 20 jobs:", [x.expanded for x in r])
eng.session_end()
say("session ended", eng.stats())
eng.close()
say("closed")
