"""The oracle against every known-answer assertion the reference's own tests hold for this path.

These are the only numeric pins the reference ships; the fixtures in tests/golden/ref_tests.json are the reference's
test DATA files (test/*.yaml, test/map_3x3.txt) plus the asserted values with their file:line.
"""
import hashlib

import numpy as np
import pytest


def test_a_star_map_3x3(oracle_mod, ref_tests):  # test/test_a_star.py:20-38 (and test_a_star_epsilon.py:21-39, w=1)
    m = ref_tests["map_3x3"]
    assert (m["dimx"], m["dimy"]) == (3, 3)
    mask = np.asarray(m["mask"], dtype=np.uint8).ravel()
    for case in m["cases"]:
        states, _ = oracle_mod.astar_2d(m["dimx"], m["dimy"], mask, case["start"], case["goal"])
        assert len(states) == case["n_states"], case


@pytest.mark.parametrize("name", ["mapf_simple1", "mapf_circle", "mapf_atGoal"])
def test_cbs_cost(oracle_mod, ref_tests, name):  # test/test_cbs.py:24-34
    r = oracle_mod.mapf_solve(oracle_mod.CBS, ref_tests["mapf"][name])
    assert r["rc"] == 1 and r["cost"] == ref_tests["cbs_cost"][name]


@pytest.mark.parametrize("name", ["mapf_simple1", "mapf_circle", "mapf_atGoal"])
def test_ecbs_w1_cost(oracle_mod, ref_tests, name):  # test/test_ecbs.py:25-35
    r = oracle_mod.mapf_solve(oracle_mod.ECBS, ref_tests["mapf"][name], w=1.0)
    assert r["rc"] == 1 and r["cost"] == ref_tests["ecbs_w1_cost"][name]


def test_sipp_1(oracle_mod, ref_tests):  # test/test_sipp.py:16-21
    s = ref_tests["sipp_1"]
    states, _ = oracle_mod.sipp_single(s["dimx"], s["dimy"], s["obstacles"], s["start"], s["goal"],
                                       s["collision_intervals"])
    assert len(states) == s["n_states"]
    assert states[-1] == s["last"]


def test_prioritized_sipp(oracle_mod, ref_tests):  # test/test_mapf_prioritized_sipp.py:24-52
    exp = ref_tests["prioritized_sipp"]
    for name, cost in exp["cost"].items():
        r = oracle_mod.prioritized_sipp(ref_tests["mapf"][name])
        assert r["cost"] == cost, name
    r = oracle_mod.prioritized_sipp(ref_tests["mapf"]["mapf_simple1b"])
    assert len(r["schedules"][0]) == exp["simple1b_lens"]["agent0"]
    assert len(r["schedules"][1]) == exp["simple1b_lens"]["agent1"]


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(("|" + ",".join("%d:%d" % (x, y) for x, y in p)).encode())
    return h.hexdigest()[:16]


def test_oracle_regression_vectors(oracle_mod, bench_instances, oracle_expected):
    """oracle_expected.json was produced by this same oracle (NOT by the reference): guards against drift."""
    names = [n for n in sorted(bench_instances) if "agents10_" in n and "32by32" in n][:25]
    names += [n for n in sorted(bench_instances) if "8by8" in n and "agents4_" in n]
    for name in names:
        inst = bench_instances[name]
        for key, exp in oracle_expected[name].items():
            if exp["rc"] != 1 or key == "ecbs_w1.0":
                continue
            if key == "cbs":
                r = oracle_mod.mapf_solve(oracle_mod.CBS, inst, cap_total=300_000)
            else:
                r = oracle_mod.mapf_solve(oracle_mod.ECBS, inst, w=float(key.split("w")[1]), cap_total=3_000_000)
            assert (r["cost"], r["makespan"], r["hl_expanded"], r["ll_expanded"]) == (
                exp["cost"], exp["makespan"], exp["hl"], exp["ll"]), (name, key)
            assert _digest(r["paths"]) == exp["digest"]


def test_cbs_sum_of_costs_is_optimal_vs_ecbs_bound(oracle_mod, bench_instances, oracle_expected):
    """Domain property (doc/libMultiRobotPlanning.md:18-23): ECBS(w) cost <= w * CBS optimal cost."""
    for name, rec in oracle_expected.items():
        if "cbs" in rec and rec["cbs"]["rc"] == 1 and rec["ecbs_w1.3"]["rc"] == 1:
            assert rec["ecbs_w1.3"]["cost"] >= rec["cbs"]["cost"]
            assert rec["ecbs_w1.3"]["cost"] <= 1.3 * rec["cbs"]["cost"] + 1e-6
