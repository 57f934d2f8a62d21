"""Command-line front-ends with the reference's flags and YAML schemas, running on the MI355X engine.

    python -m libmultirobotplanning_amd.cli ecbs -i in.yaml -o out.yaml -w 1.3     (example/ecbs.cpp:524-623)
    python -m libmultirobotplanning_amd.cli cbs  -i in.yaml -o out.yaml            (example/cbs.cpp:571-667)

Input  (ecbs.cpp:554-574):  map.dimensions [x, y], map.obstacles [[x, y]..], agents[].{name,start,goal}
Output (ecbs.cpp:584-617):  statistics.{cost, makespan, runtime, highLevelExpanded, lowLevelExpanded} and
                            schedule.agent<k>: [{x, y, t}..] — the schema example/visualize.py:63,103 consumes.
On failure the reference prints "Planning NOT successful!" and writes nothing (ecbs.cpp:618-620); so does this tool.
Several input files may be given (-i a.yaml -i b.yaml … with matching -o): they are solved as one GPU batch.
"""
import argparse
import sys
from typing import Dict, List

import yaml

from . import hl


def read_instance(path: str) -> Dict:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    dim = cfg["map"]["dimensions"]
    return dict(dimx=int(dim[0]), dimy=int(dim[1]),
                obstacles=[[int(o[0]), int(o[1])] for o in (cfg["map"].get("obstacles") or [])],
                starts=[[int(a["start"][0]), int(a["start"][1])] for a in cfg["agents"]],
                goals=[[int(a["goal"][0]), int(a["goal"][1])] for a in cfg["agents"]])


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


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="libmultirobotplanning_amd.cli")
    ap.add_argument("algo", choices=["ecbs", "cbs"])
    ap.add_argument("-i", "--input", action="append", required=True, help="input file (YAML)")
    ap.add_argument("-o", "--output", action="append", required=True, help="output file (YAML)")
    ap.add_argument("-w", "--suboptimality", type=float, default=1.0, help="suboptimality bound (ecbs)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-ll-expansions", type=int, default=-1, help="harness cap per instance (reference: none)")
    args = ap.parse_args(argv)
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


if __name__ == "__main__":
    sys.exit(main())
