import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_MEASURED = {}


@pytest.fixture(scope="session")
def measured():
    """measured(name, value): record an error figure a GPU parity test measured (kept next to its threshold in DESIGN.md §4).  The
    session writes them to gpurun_out/measured_errors.json so thresholds can be set at <= 1.3x what the hardware produced."""
    def rec(name, value):
        _MEASURED[name] = float(value)
        print(f"[measured] {name} = {float(value):.6g}")
    return rec


def pytest_sessionfinish(session, exitstatus):
    if _MEASURED:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "measured_errors.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(_MEASURED)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
