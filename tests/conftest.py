import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def pytest_sessionfinish(session, exitstatus):
    """FE_RECORD_PARITY=<file>: the largest relative rms error every GPU parity test saw (tests/test_gpu_parity.py::OBSERVED)"""
    path = os.environ.get("FE_RECORD_PARITY")
    mod = sys.modules.get("test_gpu_parity")
    if path and mod is not None and getattr(mod, "OBSERVED", None):
        import json
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump({k: {"rel_rms": v[0], "what": v[1]} for k, v in sorted(mod.OBSERVED.items())}, f, indent=1)
