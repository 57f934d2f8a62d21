import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_tests():
    with open(os.path.join(GOLDEN, "ref_tests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bench_instances():
    with open(os.path.join(GOLDEN, "bench_instances.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_expected():
    with open(os.path.join(GOLDEN, "oracle_expected.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def shipped_corpus():
    """(name, instance) pairs of all 1000 shipped benchmark/32x32_obst204 inputs + the oracle's golden results."""
    from libmultirobotplanning_amd import hl
    with open(os.path.join(GOLDEN, "shipped_32x32_expected.json")) as f:
        exp = json.load(f)
    return hl.load_shipped_corpus(os.path.join(GOLDEN, "shipped_32x32.npz")), exp


@pytest.fixture(scope="session")
def ll_jobs_golden():
    with open(os.path.join(GOLDEN, "ll_jobs.json")) as f:
        return json.load(f)
