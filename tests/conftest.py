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
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    import json
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="session")
def join_answers():
    """What the REFERENCE's own joinCpu (hjcp.cu:2013-2059, compiled from where it lies by oracle/Makefile)
    printed for pairs of golden relations: s = matching pairs, g = sum of matching S keys mod 2^32."""
    import json
    return json.load(open(os.path.join(GOLDEN, "join_answers.json")))["joins"]
