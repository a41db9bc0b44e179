import os
import sys

if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None:  # before the HIP runtime reads its flags: sihl_amd/__init__.py
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    _t = sys.modules.get("torch")
    os.environ["SIHL_GRAPH_ENV_EARLY"] = "0" if (_t is not None and _t.cuda.is_initialized()) else "1"
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
