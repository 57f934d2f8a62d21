"""Per-expansion latency of the compact tier against the number of resident wavefronts (dev tool, GPU box):
harvested ECBS low-level searches in batch mode with `slots` workgroups resident; the kernel's own 100 MHz tick counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from libmultirobotplanning_amd import ll, hl
agents = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 256
slot_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,256,1024,1792").split(",")]
jobs_src = []
for k in range(n_inst):
    inst = hl.generate_instance(1000 * agents + k, 32, 32, 204, agents)
    summary, calls = oracle.mapf_record(oracle.ECBS, inst, w=1.3)
    jobs_src.append((inst, calls))
for slots in slot_list:
    eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=slots)
    jobs, exp = [], []
    for inst, calls in jobs_src:
        mid = eng.upload_map(inst["dimx"], inst["dimy"], inst["obstacles"])
        for c in calls:
            jobs.append(ll.LLJob(map_id=mid, algo=ll.ASTAR_EPS, start=inst["starts"][c["agent"]], goal=inst["goals"][c["agent"]],
                                 agent_idx=c["agent"], w=1.3, vertex_constraints=c["vertex_constraints"],
                                 edge_constraints=c["edge_constraints"], ctx_paths=c["ctx_paths"]))
            exp.append(c["expanded"])
    if slots <= 4:
        jobs, exp = jobs[:400], exp[:400]
    eng.search_batch(jobs[:64])
    eng.reset_stats()
    t0 = time.time()
    res = eng.search_batch(jobs)
    dt = time.time() - t0
    st = eng.stats()
    assert [r.expanded for r in res] == exp or os.environ.get('MRP_OCC_NO_ASSERT')
    p = st["prof"]
    print("slots %5d: jobs %d expansions %d wall %.1f ms kernel %.2f ms | compact %.3f us/expansion over %d, arena %.3f us/expansion over %d, job %.1f us, handed over %d" % (
        slots, len(jobs), sum(exp), dt * 1e3, st["kernel_ms"], p[0] / 100.0 / max(p[1], 1), p[1], p[2] / 100.0 / max(p[3], 1), p[3],
        p[4] / 100.0 / max(p[5], 1), p[7]), flush=True)
    eng.close()
