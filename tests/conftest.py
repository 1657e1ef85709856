import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _native_backtrace_on_fault():
    """HJ_TEST_SEGV_BT=<path of tools/segv_bt.c built as a shared library>: (re-)install its fault handler before every test — a native
    backtrace of the faulting thread where pytest's faulthandler has Python frames only (tools/gpu_dist_soak.sh)."""
    lib = os.environ.get("HJ_TEST_SEGV_BT")
    if lib:
        import ctypes
        ctypes.CDLL(lib).segv_bt_install()
    yield


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
