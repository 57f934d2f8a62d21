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

* ``shipped_32x32.npz``     — ALL 1000 shipped benchmark/32x32_obst204 inputs (agents10..100 x ex0..99) as uint8 arrays
                              (obstacles, starts, goals per agent count): the corpus BASELINE.json's north_star names.
* ``shipped_32x32_expected.json`` — our oracle's ECBS w=1.3 results on every one of them at a cap of 3 000 000 low-level
                              expansions per instance (agents100_ex36 runs into it), same status as oracle_expected.json.
* ``cbs_8x8_cap1e6.json``   — our oracle's CBS results on shipped 8x8_obst12 inputs at SURVEY.md §8(d)(iii)'s cap of
                              1 000 000 low-level expansions per instance (agents 10, 12, 16, 20).
* ``shipped_heavy_tail_expected.json`` — our oracle's result for agents100_ex36 with NO cap (the reference has none).
* ``ll_jobs.json``          — ~200 low-level searches harvested from the oracle's conflict trees (inputs AND outputs), so
                              that a box without a compiler for the oracle still has low-level parity vectors.

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
    t["cbs_ta"] = ta_tests()
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


def shipped_corpus():
    """All 1000 shipped 32x32_obst204 instances, compactly, + oracle vectors at cap 3e6 (a process pool: ~2 CPU-minutes)."""
    import numpy as np
    from concurrent.futures import ProcessPoolExecutor
    arrays = {}
    jobs = []
    for n in range(10, 101, 10):
        obst = np.zeros((100, 204, 2), dtype=np.uint8)
        starts = np.zeros((100, n, 2), dtype=np.uint8)
        goals = np.zeros((100, n, 2), dtype=np.uint8)
        for k in range(100):
            inst = load_mapf_yaml(os.path.join(REF, "benchmark", "32x32_obst204", "map_32by32_obst204_agents%d_ex%d.yaml" % (n, k)))
            assert inst["dimx"] == 32 and inst["dimy"] == 32 and len(inst["obstacles"]) == 204 and len(inst["starts"]) == n
            obst[k] = inst["obstacles"]
            starts[k] = inst["starts"]
            goals[k] = inst["goals"]
            jobs.append(("map_32by32_obst204_agents%d_ex%d" % (n, k), inst))
        arrays["obst%d" % n] = obst
        arrays["starts%d" % n] = starts
        arrays["goals%d" % n] = goals
    np.savez_compressed(os.path.join(OUT, "shipped_32x32.npz"), **arrays)
    jobs.sort(key=lambda j: -len(j[1]["starts"]))  # the heavy ones first
    with ProcessPoolExecutor(max_workers=7) as pool:
        res = list(pool.map(_solve_shipped, jobs, chunksize=4))
    exp = dict(res)
    with open(os.path.join(OUT, "shipped_32x32_expected.json"), "w") as f:
        json.dump(exp, f, separators=(",", ":"), sort_keys=True)
    return exp


def _solve_shipped(job):
    import oracle
    name, inst = job
    r = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=3_000_000, path_cap=1024)
    return name, summarize(r)


def _solve_cbs_1e6(job):
    import oracle
    name, inst = job
    return name, summarize(oracle.mapf_solve(oracle.CBS, inst, cap_total=1_000_000, path_cap=1024))


def cbs_8x8_cap1e6():
    from concurrent.futures import ProcessPoolExecutor
    jobs = []
    for n in (10, 12, 16, 20):
        for k in range(4):
            name = "map_8by8_obst12_agents%d_ex%d" % (n, k)
            jobs.append((name, load_mapf_yaml(os.path.join(REF, "benchmark", "8x8_obst12", name + ".yaml"))))
    with ProcessPoolExecutor(max_workers=7) as pool:
        exp = dict(pool.map(_solve_cbs_1e6, jobs))
    with open(os.path.join(OUT, "cbs_8x8_cap1e6.json"), "w") as f:
        json.dump(exp, f, separators=(",", ":"), sort_keys=True)
    return exp


def ll_jobs(instances):
    """~200 low-level calls with their oracle results: 32x32 ECBS w=1.3 (agents 10..50), 8x8 CBS and ECBS w=1."""
    import oracle
    picks = [("map_32by32_obst204_agents10_ex%d" % k, oracle.ECBS, 1.3, 8) for k in range(0, 60, 4)]
    picks += [("map_32by32_obst204_agents20_ex%d" % k, oracle.ECBS, 1.3, 6) for k in range(4)]
    picks += [("map_32by32_obst204_agents30_ex%d" % k, oracle.ECBS, 1.3, 4) for k in range(3)]
    picks += [("map_32by32_obst204_agents50_ex%d" % k, oracle.ECBS, 1.3, 4) for k in range(3)]
    picks += [("map_8by8_obst12_agents%d_ex%d" % (n, k), oracle.CBS, 1.0, 4) for n in (4, 6, 8) for k in range(2)]
    picks += [("map_8by8_obst12_agents%d_ex%d" % (n, k), oracle.ECBS, 1.0, 3) for n in (5, 8) for k in range(2)]
    out = []
    for name, algo, w, take in picks:
        inst = instances[name]
        _, calls = oracle.mapf_record(algo, inst, w=w, cap_total=300_000)
        # the last calls of a tree carry the most constraints; keep a spread
        idx = sorted(set([0, len(calls) - 1] + [int(i * (len(calls) - 1) / max(take - 1, 1)) for i in range(take)]))[:take]
        for i in idx:
            c = calls[i]
            out.append(dict(instance=name, algo="ecbs" if algo == oracle.ECBS else "cbs", w=w, agent=c["agent"],
                            vertex_constraints=c["vertex_constraints"], edge_constraints=c["edge_constraints"],
                            ctx_paths=c["ctx_paths"] if algo == oracle.ECBS else [], success=c["success"], cost=c["cost"],
                            fmin=c["fmin"], expanded=c["expanded"], states=c["states"]))
    with open(os.path.join(OUT, "ll_jobs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    return out


def ta_tests():
    """test/mapfta_simple1_a{1,2,3}.yaml (data) + what test/test_cbs_ta.py:24-38 asserts about them."""
    out = {}
    for k in (1, 2, 3):
        with open(os.path.join(REF, "test", "mapfta_simple1_a%d.yaml" % k)) as f:
            cfg = yaml.safe_load(f)
        dim = cfg["map"]["dimensions"]
        out["mapfta_simple1_a%d" % k] = dict(
            dimx=int(dim[0]), dimy=int(dim[1]), obstacles=[[int(o[0]), int(o[1])] for o in (cfg["map"]["obstacles"] or [])],
            starts=[[int(a["start"][0]), int(a["start"][1])] for a in cfg["agents"]],
            potential_goals=[[[int(g[0]), int(g[1])] for g in (a["potentialGoals"] or [])] for a in cfg["agents"]])
    return dict(inputs=out, cost=dict(mapfta_simple1_a1=6, mapfta_simple1_a2=6, mapfta_simple1_a3=5),
                ends=dict(mapfta_simple1_a2=dict(agent0=[4, 0, 4], agent1_xy=[2, 1]), mapfta_simple1_a3=dict(agent0=[3, 0, 3])),
                src="test/test_cbs_ta.py:24-38 (agent end states as x, y, t)")


def heavy_tail():
    """The one shipped input the 3 000 000-expansion cap of shipped_32x32_expected.json cuts off — agents100_ex36 — run to
    completion by our oracle (no cap at all, as the reference has none): SURVEY.md §6 reports cost 2574, 70 612 conflict-tree
    nodes and 56 795 846 low-level expansions from the reference's own headers."""
    import time
    import numpy as np
    import oracle
    oracle.build()
    z = np.load(os.path.join(OUT, "shipped_32x32.npz"))
    inst = dict(dimx=32, dimy=32, obstacles=z["obst100"][36].astype(int).tolist(), starts=z["starts100"][36].astype(int).tolist(),
                goals=z["goals100"][36].astype(int).tolist())
    t0 = time.time()
    r = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=-1, path_cap=1024)
    rec = summarize(r)
    rec["oracle_search_seconds_in_the_build_container"] = r["elapsed_ns"] / 1e9
    rec["wall_seconds_in_the_build_container"] = time.time() - t0
    with open(os.path.join(OUT, "shipped_heavy_tail_expected.json"), "w") as f:
        json.dump({"map_32by32_obst204_agents100_ex36": rec}, f, separators=(",", ":"), sort_keys=True)
    return rec


if __name__ == "__main__":
    if "--heavy-tail" in sys.argv:  # round-4 addition: agents100_ex36 uncapped
        print(heavy_tail())
        sys.exit(0)
    if "--ta" in sys.argv:  # round-3 addition: the task-assignment fixtures go into ref_tests.json
        with open(os.path.join(OUT, "ref_tests.json")) as f:
            t = json.load(f)
        t["cbs_ta"] = ta_tests()
        with open(os.path.join(OUT, "ref_tests.json"), "w") as f:
            json.dump(t, f, separators=(",", ":"))
        sys.exit(0)
    if "--shipped" in sys.argv:  # only the round-3 additions (the other files are unchanged)
        with open(os.path.join(OUT, "bench_instances.json")) as f:
            inst = json.load(f)
        print("ll jobs", len(ll_jobs(inst)))
        print("cbs 8x8 cap 1e6", cbs_8x8_cap1e6())
        exp = shipped_corpus()
        print("shipped: %d instances, %d capped" % (len(exp), sum(1 for v in exp.values() if v["rc"] != 1)))
        sys.exit(0)
    with open(os.path.join(OUT, "ref_tests.json"), "w") as f:
        json.dump(ref_tests(), f, separators=(",", ":"))
    inst = bench_instances()
    with open(os.path.join(OUT, "bench_instances.json"), "w") as f:
        json.dump(inst, f, separators=(",", ":"))
    exp = oracle_expected(inst)
    with open(os.path.join(OUT, "oracle_expected.json"), "w") as f:
        json.dump(exp, f, separators=(",", ":"), sort_keys=True)
