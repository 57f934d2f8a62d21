"""Effect of the conflict-tree look-ahead (MRP_HL_SPEC) on small batches: the 150 shipped 32x32 inputs (ECBS w=1.3), the 10
shipped agents100 inputs alone, and CBS on the shipped 8x8 agents8 inputs.  Prints wall time per width; results are
checked to be identical across widths."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import hl
inst = json.load(open(os.path.join(ROOT, "tests/golden/bench_instances.json")))
sets = {
    "shipped150_ecbs": ([inst[n] for n in sorted(inst) if "32by32" in n], hl.ECBS, 3000000),
    "agents100x10_ecbs": ([inst[n] for n in sorted(inst) if "agents100_" in n], hl.ECBS, 3000000),
    "cbs_8x8_agents8": ([inst[n] for n in sorted(inst) if "8by8_obst12_agents8_" in n], hl.CBS, 300000),
    "cbs_8x8_agents10_12": ([inst[n] for n in sorted(inst) if "8by8_obst12_agents10_" in n or "8by8_obst12_agents12_" in n], hl.CBS, 300000),
}
s = hl.BatchSolver(device=0, n_threads=int(os.environ.get("THREADS", "16")), slots=512)
for name, (insts, algo, cap) in sets.items():
    ref = None
    for spec in (1, 2, 4, 8, 16):
        os.environ["MRP_HL_SPEC"] = str(spec)
        best = None
        for rep in range(3):
            res, st = s.solve(insts, algo=algo, w=1.3, max_ll_expansions=cap, want_paths=False)
            if best is None or st["wall_seconds"] < best["wall_seconds"]:
                best = st
        # a capped instance's expansion total depends on how far its last searches got; everything else is width-independent
        key = [(r["status"], r["cost"], r["hl_expanded"], r["ll_expanded"] if r["status"] == hl.SOLVED else -1) for r in res]
        if ref is None:
            ref = key
        assert key == ref, (name, spec)
        print("%-22s spec %2d: wall %.4f s  exp %d (%.3e/s)  wasted %d  ahead-searches %d  searches %d" % (
            name, spec, best["wall_seconds"], best["ll_expansions"], best["ll_expansions"] / best["wall_seconds"],
            best["wasted_ll_expansions"], best["speculative_searches"], best["ll_searches"]), flush=True)
s.close()
