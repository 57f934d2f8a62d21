#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

What is produced, and from what:

* ``ref_tests.json``        — the reference's own test inputs (test/*.yaml, test/map_3x3.txt: DATA files, parsed with
                              yaml.safe_load / plain text) together with the expected values its own unittest scripts
                              assert (test/test_a_star.py, test_a_star_epsilon.py, test_cbs.py, test_ecbs.py,
                              test_sipp.py, test_mapf_prioritized_sipp.py — the file:line of each assertion is recorded).
                              These are the only numeric pins the reference ships.
* ``bench_instances.json``  — a compact copy of shipped benchmark INPUT instances (benchmark/32x32_obst204,
                              benchmark/8x8_obst12): obstacles, starts, goals.  Inputs only.
* ``oracle_expected.json``  — outputs of OUR oracle (oracle/liboracle.so) on those instances: cost, makespan,
                              high/low-level expansion counts, path checksum.  These are regression vectors for the
                              HIP path, NOT reference outputs: the reference is unbuildable here (no Boost / yaml-cpp),
                              see DESIGN.md "Oracle pinning".

No reference source text is copied; only data files are converted.
"""
import hashlib
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_mapf_yaml(path):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    dim = cfg["map"]["dimensions"]
    return dict(dimx=int(dim[0]), dimy=int(dim[1]),
                obstacles=[[int(o[0]), int(o[1])] for o in (cfg["map"]["obstacles"] or [])],
                starts=[[int(a["start"][0]), int(a["start"][1])] for a in cfg["agents"]],
                goals=[[int(a["goal"][0]), int(a["goal"][1])] for a in cfg["agents"]])


def path_digest(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(("|" + ",".join("%d:%d" % (x, y) for x, y in p)).encode())
    return h.hexdigest()[:16]


def ref_tests():
    t = {}
    # --- a_star / a_star_epsilon on the 3x3 text map (a_star.cpp:163-185: '#' = obstacle, dimY = lines-1) ---
    with open(os.path.join(REF, "test/map_3x3.txt")) as f:
        text = f.read()
    lines = text.split("\n")          # getline loop incl. the empty line after the final newline
    dimx = max(len(l) for l in lines)
    dimy = len(lines) - 1
    mask = [[1 if (x < len(lines[y]) and lines[y][x] == "#") else 0 for x in range(dimx)] for y in range(dimy)]
    t["map_3x3"] = dict(dimx=dimx, dimy=dimy, mask=mask, cases=[
        dict(start=[0, 0], goal=[0, 0], n_states=1, src="test/test_a_star.py:20-22"),
        dict(start=[0, 0], goal=[1, 1], n_states=0, src="test/test_a_star.py:24-26 (output None)"),
        dict(start=[1, 1], goal=[0, 0], n_states=0, src="test/test_a_star.py:28-30"),
        dict(start=[1, 1], goal=[2, 2], n_states=0, src="test/test_a_star.py:32-34"),
        dict(start=[0, 0], goal=[2, 1], n_states=4, src="test/test_a_star.py:36-38"),
    ])
    # --- MAPF yaml fixtures ---
    mapf = {}
    for name in ["mapf_simple1", "mapf_circle", "mapf_atGoal", "mapf_simple1b", "mapf_someAtGoal", "mapf_swap2",
                 "mapf_swap4"]:
        mapf[name] = load_mapf_yaml(os.path.join(REF, "test", name + ".yaml"))
    t["mapf"] = mapf
    t["cbs_cost"] = dict(mapf_simple1=8, mapf_circle=4, mapf_atGoal=0, src="test/test_cbs.py:24-34")
    t["ecbs_w1_cost"] = dict(mapf_simple1=8, mapf_circle=4, mapf_atGoal=0, src="test/test_ecbs.py:25-35")
    t["prioritized_sipp"] = dict(
        cost=dict(mapf_simple1=8, mapf_simple1b=2, mapf_circle=4, mapf_atGoal=0, mapf_swap2=12, mapf_swap4=28,
                  mapf_someAtGoal=0),
        simple1b_lens=dict(agent0=3, agent1=0),
        src="test/test_mapf_prioritized_sipp.py:24-52")
    with open(os.path.join(REF, "test/sipp_1.yaml")) as f:
        s = yaml.safe_load(f)
    ci = []
    for node in s["environment"]["collisionIntervals"]:
        for iv in node["intervals"]:
            ci.append([int(node["location"][0]), int(node["location"][1]), int(iv[0]), int(iv[1])])
    t["sipp_1"] = dict(dimx=int(s["environment"]["size"][0]), dimy=int(s["environment"]["size"][1]),
                       obstacles=[[int(o[0]), int(o[1])] for o in (s["environment"]["obstacles"] or [])],
                       start=[int(v) for v in s["start"]], goal=[int(v) for v in s["goal"]],
                       collision_intervals=ci, n_states=6, last=[2, 3, 9], src="test/test_sipp.py:16-21")
    return t


def bench_instances():
    out = {}
    sel = [("32x32_obst204", "map_32by32_obst204_agents%d_ex%d.yaml", n, range(cnt))
           for n, cnt in [(10, 100), (20, 10), (30, 10), (50, 20), (100, 10)]]
    sel += [("8x8_obst12", "map_8by8_obst12_agents%d_ex%d.yaml", n, range(cnt))
            for n, cnt in [(2, 10), (4, 10), (5, 10), (6, 10), (8, 10), (10, 4), (12, 4), (16, 4), (20, 4)]]
    for d, pat, n, rng in sel:
        for k in rng:
            name = pat % (n, k)
            inst = load_mapf_yaml(os.path.join(REF, "benchmark", d, name))
            out[name[:-5]] = inst
    return out


def oracle_expected(instances):
    import oracle
    oracle.build()
    exp = {}
    for name, inst in instances.items():
        n = len(inst["starts"])
        rec = {}
        if "32by32" in name:
            for w in (1.3,) if n > 10 else (1.0, 1.3):
                r = oracle.mapf_solve(oracle.ECBS, inst, w=w, cap_total=3_000_000)
                rec["ecbs_w%.1f" % w] = summarize(r)
        else:
            r = oracle.mapf_solve(oracle.CBS, inst, cap_total=300_000)
            rec["cbs"] = summarize(r)
            r = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=300_000)
            rec["ecbs_w1.3"] = summarize(r)
        exp[name] = rec
        print(name, rec, flush=True)
    return exp


def summarize(r):
    if r["rc"] != 1:
        return dict(rc=r["rc"])
    return dict(rc=1, cost=r["cost"], makespan=r["makespan"], hl=r["hl_expanded"], ll=r["ll_expanded"],
                digest=path_digest(r["paths"]))


if __name__ == "__main__":
    with open(os.path.join(OUT, "ref_tests.json"), "w") as f:
        json.dump(ref_tests(), f, separators=(",", ":"))
    inst = bench_instances()
    with open(os.path.join(OUT, "bench_instances.json"), "w") as f:
        json.dump(inst, f, separators=(",", ":"))
    exp = oracle_expected(inst)
    with open(os.path.join(OUT, "oracle_expected.json"), "w") as f:
        json.dump(exp, f, separators=(",", ":"), sort_keys=True)
