"""Front-end pieces that need no GPU: the dependency-free YAML-subset reader (yaml_subset.py) against PyYAML, and the
`a_star` front-end (BASELINE.json configs[0]: host plumbing) against the reference's own test cases."""
import random

import pytest
import yaml


def test_yaml_subset_matches_pyyaml():
    from libmultirobotplanning_amd import yaml_subset
    rng = random.Random(5)
    docs = []
    for _ in range(60):
        n = rng.randrange(0, 6)
        docs.append({"map": {"dimensions": [rng.randrange(1, 99), rng.randrange(1, 99)],
                             "obstacles": [[rng.randrange(50), rng.randrange(50)] for _ in range(rng.randrange(0, 9))]},
                     "agents": [{"name": "agent%d" % i, "start": [rng.randrange(9), rng.randrange(9)],
                                 "goal": [rng.randrange(9), rng.randrange(9)]} for i in range(n)]})
        docs.append({"start": [0, 1], "goal": [2, 3],
                     "environment": {"size": [10, 10], "obstacles": [],
                                     "collisionIntervals": [{"location": [rng.randrange(9), 3],
                                                             "intervals": [[0, 8], [12, rng.randrange(13, 30)]]}
                                                            for _ in range(rng.randrange(0, 4))]}})
    for d in docs:
        for style in (None, False, True):
            text = yaml.safe_dump(d, default_flow_style=style)
            if style is True:
                continue  # flow MAPPINGS are outside the subset (the reference's files never use them)
            assert yaml_subset.loads(text) == yaml.safe_load(text), text
    # the two layouts the reference's files use (benchmark/*.yaml: PyYAML block style with 4-space "-   key"; test/*.yaml)
    bench = "agents:\n-   goal: [11, 20]\n    name: agent0\n    start: [4, 21]\n-   goal: [6, 5]\n    name: agent1\n" \
            "    start: [20, 4]\nmap:\n    dimensions: [32, 32]\n    obstacles:\n    - [14, 22]\n    - [30, 24]\n"
    hand = "map:\n  dimensions: [5, 2]\n  obstacles:\n    - [0, 1]\n    - [1, 1]  # comment\nagents:\n  - name: agent0\n" \
           "    start: [0, 0]\n    goal: [4, 0]\n"
    for text in (bench, hand, "a: []\nb:\nc: 1.5\n", "- [1, [2, 3]]\n- x\n"):
        assert yaml_subset.loads(text) == yaml.safe_load(text)
    for bad in ("a: {b: 1}\n", "a: &x 1\n", "a: [1, 2\n", "a: 1\n  b: 2\n", "\ta: 1\n", "a: [1,,2]\n"):
        with pytest.raises(ValueError):
            yaml_subset.loads(bad)


def test_a_star_front_end(tmp_path, ref_tests, capsys):
    """example/a_star.cpp on test/map_3x3.txt's content: the cases of test/test_a_star.py:20-38."""
    from libmultirobotplanning_amd import cli
    m = ref_tests["map_3x3"]
    text = "".join("".join("#" if c else "." for c in row) + "\n" for row in m["mask"])
    (tmp_path / "map.txt").write_text(text)
    for c in m["cases"]:
        out = tmp_path / "out.yaml"
        assert cli.main(["a_star", "--startX", str(c["start"][0]), "--startY", str(c["start"][1]), "--goalX",
                         str(c["goal"][0]), "--goalY", str(c["goal"][1]), "-m", str(tmp_path / "map.txt"), "-o",
                         str(out)]) == 0
        doc = yaml.safe_load(open(out))
        printed = capsys.readouterr().out
        if c["n_states"] == 0:
            assert doc is None and "Planning NOT successful!" in printed, c["src"]
        else:
            steps = doc["schedule"]["agent1"]
            assert len(steps) == c["n_states"], c["src"]
            assert [s["t"] for s in steps] == list(range(len(steps)))
            assert [steps[0]["x"], steps[0]["y"]] == c["start"] and [steps[-1]["x"], steps[-1]["y"]] == c["goal"]
            assert "Planning successful! Total cost: %d" % (len(steps) - 1) in printed


def test_host_a_star_matches_oracle_on_random_maps(oracle_mod):
    from libmultirobotplanning_amd import hl
    rng = random.Random(3)
    for _ in range(1500):
        dx, dy = rng.randrange(2, 14), rng.randrange(2, 14)
        m = [[1 if rng.random() < 0.25 else 0 for _ in range(dx)] for _ in range(dy)]
        s = [rng.randrange(dx), rng.randrange(dy)]
        g = [rng.randrange(dx), rng.randrange(dy)]
        st, cost, exp = hl.astar_grid2d(dx, dy, m, s, g)
        o, oe = oracle_mod.astar_2d(dx, dy, m, s, g)
        if m[s[1]][s[0]]:
            o = []  # a_star.cpp:190 only searches from a valid start
        assert st == o
        if o:
            assert (cost, exp) == (len(o) - 1, oe)
