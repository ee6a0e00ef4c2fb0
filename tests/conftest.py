import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# every parity test records the rel-L2 (or max-abs) it measured next to the tolerance it asserts; on the GPU box the file lands under
# gpurun_out/ (merged back by gpurun) and the per-round copy is committed as profiles/rNN_parity_report.json
PARITY_REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.json")
_PARITY = {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:      # keep `@pytest.mark.timeout` a known marker when the plugin is missing (it is installed in this image)
        config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout)")


class Measured(float):
    """A measured parity figure: `assert measured(x) < tol` records (x, tol) under the running test in the parity report."""

    def __lt__(self, tol):
        node = os.environ.get("PYTEST_CURRENT_TEST", "?").rsplit(" (", 1)[0]
        _PARITY.setdefault(node, []).append({"what": "", "measured": float(self), "tol": float(tol)})
        return float(self) < float(tol)


def measured(x):
    return Measured(float(x))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def parity(request):
    """parity(measured, tol, what="") -> asserts measured < tol and appends {test, what, measured, tol} to the parity report."""
    def rec(measured, tol, what=""):
        measured = float(measured)
        _PARITY.setdefault(request.node.nodeid, []).append({"what": what, "measured": measured, "tol": float(tol)})
        assert measured < tol, f"{what or request.node.name}: measured {measured:.4e} >= tolerance {tol:.1e}"
        return measured
    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    try:
        os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
        old = {}
        if os.path.exists(PARITY_REPORT):
            with open(PARITY_REPORT) as f:
                old = json.load(f)
        old.update(_PARITY)
        with open(PARITY_REPORT, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


def free_port():
    """A TCP port nobody listens on right now (bound on 127.0.0.1 and released): rendezvous port of the multi-process gloo tests."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
