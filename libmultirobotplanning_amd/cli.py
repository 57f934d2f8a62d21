"""Command-line front-ends with the reference's flags, input formats and output schemas.

    python -m libmultirobotplanning_amd.cli ecbs -i in.yaml -o out.yaml -w 1.3     (example/ecbs.cpp:524-623)
    python -m libmultirobotplanning_amd.cli cbs  -i in.yaml -o out.yaml            (example/cbs.cpp:571-667)
    python -m libmultirobotplanning_amd.cli sipp -i in.yaml -o out.yaml            (example/sipp.cpp:149-236)
    python -m libmultirobotplanning_amd.cli mapf_prioritized_sipp -i in.yaml -o out.yaml
                                                                                   (example/mapf_prioritized_sipp.cpp:157-295)
    python -m libmultirobotplanning_amd.cli a_star --startX 0 --startY 0 --goalX 2 --goalY 1 -m map.txt -o out.yaml
                                                                                   (example/a_star.cpp:128-213)

ecbs / cbs   input  (ecbs.cpp:554-574): map.dimensions [x, y], map.obstacles [[x, y]..], agents[].{name,start,goal}
             output (ecbs.cpp:584-617): statistics.{cost, makespan, runtime, highLevelExpanded, lowLevelExpanded} and
                                        schedule.agent<k>: [{x, y, t}..] — what example/visualize.py:63,103 consumes.
sipp         input  (sipp.cpp:181-205): start, goal, environment.{size, obstacles, collisionIntervals[].{location,intervals}}
             output (sipp.cpp:225-232): schedule.agent1: [{x, y, t}..]
mapf_prioritized_sipp  input as ecbs; output (mapf_prioritized_sipp.cpp:211-272): schedule.agent<k> ("[]" for an agent
             that could not be planned), then statistics.cost
a_star       text map, '#' = obstacle (a_star.cpp:163-178: width = longest line, height = lines incl. the empty one after
             the last newline, minus 1); output schedule.agent1 with t = index (a_star.cpp:207-213).  Host plumbing
             (BASELINE.json configs[0]): no GPU involved.
On failure the reference prints "Planning NOT successful!" (and, for ecbs / cbs, writes nothing); so does this tool.
Several input files may be given to ecbs / cbs / mapf_prioritized_sipp (-i a.yaml -i b.yaml ... with matching -o): they
are solved as one GPU batch.  Inputs are read with the package's own YAML-subset reader (yaml_subset.py): no PyYAML.
"""
import argparse
import sys
from typing import Dict, List

from . import yaml_subset


def read_instance(path: str) -> Dict:
    cfg = yaml_subset.load(path)
    dim = cfg["map"]["dimensions"]
    return dict(dimx=int(dim[0]), dimy=int(dim[1]),
                obstacles=[[int(o[0]), int(o[1])] for o in (cfg["map"].get("obstacles") or [])],
                starts=[[int(a["start"][0]), int(a["start"][1])] for a in cfg["agents"]],
                goals=[[int(a["goal"][0]), int(a["goal"][1])] for a in cfg["agents"]])


def read_sipp(path: str) -> Dict:
    cfg = yaml_subset.load(path)
    env = cfg["environment"]
    ci = []
    for node in env.get("collisionIntervals") or []:
        for iv in node["intervals"]:
            ci.append([int(node["location"][0]), int(node["location"][1]), int(iv[0]), int(iv[1])])
    return dict(dimx=int(env["size"][0]), dimy=int(env["size"][1]),
                obstacles=[[int(o[0]), int(o[1])] for o in (env.get("obstacles") or [])],
                start=[int(v) for v in cfg["start"]], goal=[int(v) for v in cfg["goal"]], collision_intervals=ci)


def read_text_map(path: str):
    """a_star.cpp:163-178: std::getline until !good() — the empty read after the final newline counts as a line."""
    with open(path) as f:
        text = f.read()
    lines = text.split("\n")
    dimx = max(len(l) for l in lines)
    dimy = len(lines) - 1
    mask = [[1 if (x < len(lines[y]) and lines[y][x] == "#") else 0 for x in range(dimx)] for y in range(dimy)]
    return dimx, dimy, mask


def write_schedule(path: str, res: Dict, runtime: float) -> None:
    # written by hand to keep the reference's key order (ecbs.cpp:593-617)
    with open(path, "w") as out:
        out.write("statistics:\n")
        out.write("  cost: %d\n" % res["cost"])
        out.write("  makespan: %d\n" % res["makespan"])
        out.write("  runtime: %g\n" % runtime)
        out.write("  highLevelExpanded: %d\n" % res["hl_expanded"])
        out.write("  lowLevelExpanded: %d\n" % res["ll_expanded"])
        out.write("schedule:\n")
        for a, p in enumerate(res["paths"]):
            out.write("  agent%d:\n" % a)
            for t, (x, y) in enumerate(p):
                out.write("    - x: %d\n      y: %d\n      t: %d\n" % (x, y, t))


def _write_states(out, states_xyt) -> None:
    for x, y, t in states_xyt:
        out.write("    - x: %d\n      y: %d\n      t: %d\n" % (x, y, t))


def main_mapf(args, ap) -> int:
    from . import hl
    if len(args.input) != len(args.output):
        ap.error("give one -o per -i")
    insts = [read_instance(p) for p in args.input]
    solver = hl.BatchSolver(device=args.device, n_threads=min(len(insts), 16))
    try:
        res, st = solver.solve(insts, algo=hl.ECBS if args.algo == "ecbs" else hl.CBS, w=args.suboptimality,
                               max_ll_expansions=args.max_ll_expansions, path_cap=1024)
    finally:
        solver.close()
    rc = 0
    for r, out in zip(res, args.output):
        if r["status"] == hl.SOLVED:
            print("Planning successful! ")
            write_schedule(out, r, st["wall_seconds"])
        else:
            print("Planning NOT successful!" + (" (harness cap)" if r["status"] == hl.CAP else ""))
            rc = 1 if r["status"] != hl.NO_SOLUTION else rc
    return rc


def main_prioritized_sipp(args, ap) -> int:
    from . import hl
    if len(args.input) != len(args.output):
        ap.error("give one -o per -i")
    insts = [read_instance(p) for p in args.input]
    solver = hl.BatchSolver(device=args.device, n_threads=min(len(insts), 16), max_horizon=1024,
                            max_cells=max(4096, max(i["dimx"] * i["dimy"] for i in insts)))
    try:
        res, _ = solver.prioritized_sipp(insts, state_cap=2048)
    finally:
        solver.close()
    for r, path in zip(res, args.output):
        with open(path, "w") as out:  # mapf_prioritized_sipp.cpp:211-272
            out.write("schedule:\n")
            for a, sched in enumerate(r["schedules"]):
                print("Planning for agent %d" % a)
                out.write("  agent%d:\n" % a)
                if r["planned"][a]:
                    print("Planning successful! Total cost: %d" % (sched[-1][2]))
                    _write_states(out, sched)
                else:
                    print("Planning NOT successful!")
                    out.write("    []\n")
            out.write("statistics:\n  cost: %d\n" % r["cost"])
    return 0


def main_sipp(args, ap) -> int:
    from . import ll
    if len(args.input) != 1 or len(args.output) != 1:
        ap.error("sipp takes one -i and one -o")
    cfg = read_sipp(args.input[0])
    eng = ll.LowLevelEngine(device=args.device, n_tickets=1, slots=16, max_horizon=1024,
                            max_cells=max(4096, cfg["dimx"] * cfg["dimy"]))
    try:
        mid = eng.upload_map(cfg["dimx"], cfg["dimy"], cfg["obstacles"])
        r = eng.search_batch([ll.LLJob(map_id=mid, algo=ll.SIPP, start=cfg["start"], goal=cfg["goal"],
                                       collision_intervals=cfg["collision_intervals"])], states_cap=4096)[0]
    finally:
        eng.close()
    if not r.success:
        print("Planning NOT successful!")
        return 0
    print("Planning successful! Total cost: %d" % r.cost)  # sipp.cpp:213-223
    for k in range(len(r.actions)):
        t, x, y = r.states[k]
        print("%d: (%d,%d)->%s(cost: %d)" % (t, x, y, ll.ACTION_NAMES[r.actions[k]], r.action_costs[k]))
    t, x, y = r.states[-1]
    print("%d: (%d,%d)" % (t, x, y))
    with open(args.output[0], "w") as out:
        out.write("schedule:\n  agent1:\n")
        _write_states(out, [(x, y, t) for t, x, y in r.states])
    return 0


def main_a_star(args, ap) -> int:
    from . import hl
    for k in ("startX", "startY", "goalX", "goalY", "map"):
        if getattr(args, k) is None:
            ap.error("a_star needs --startX --startY --goalX --goalY -m -o")
    if len(args.output) != 1:
        ap.error("a_star takes one -o")
    dimx, dimy, mask = read_text_map(args.map)
    print("%d %d" % (dimx, dimy + 1))  # the reference prints the line count, not the height it passes on (a_star.cpp:179)
    states, cost, _ = ([], 0, 0) if dimx <= 0 or dimy <= 0 else hl.astar_grid2d(
        dimx, dimy, mask, [args.startX, args.startY], [args.goalX, args.goalY])
    with open(args.output[0], "w") as out:  # the file is created either way (a_star.cpp:194)
        if states:
            print("Planning successful! Total cost: %d" % cost)
            out.write("schedule:\n  agent1:\n")
            _write_states(out, [(x, y, t) for t, (x, y) in enumerate(states)])
        else:
            print("Planning NOT successful!")
    return 0


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="libmultirobotplanning_amd.cli")
    ap.add_argument("algo", choices=["ecbs", "cbs", "sipp", "mapf_prioritized_sipp", "a_star"])
    ap.add_argument("-i", "--input", action="append", default=[], help="input file (YAML)")
    ap.add_argument("-o", "--output", action="append", required=True, help="output file (YAML)")
    ap.add_argument("-w", "--suboptimality", type=float, default=1.0, help="suboptimality bound (ecbs)")
    ap.add_argument("-m", "--map", help="input map (txt) (a_star)")
    for k in ("startX", "startY", "goalX", "goalY"):
        ap.add_argument("--" + k, type=int, help="a_star: %s" % k)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-ll-expansions", type=int, default=-1, help="harness cap per instance (reference: none)")
    args = ap.parse_args(argv)
    if args.algo != "a_star" and not args.input:
        ap.error("-i is required")
    if args.algo in ("ecbs", "cbs"):
        return main_mapf(args, ap)
    if args.algo == "mapf_prioritized_sipp":
        return main_prioritized_sipp(args, ap)
    if args.algo == "sipp":
        return main_sipp(args, ap)
    return main_a_star(args, ap)


if __name__ == "__main__":
    sys.exit(main())
