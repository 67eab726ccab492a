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


def ts_range(tok):
    s = [v for k, v in tok.event_start.items() if k.name == "TIME_SHIFT"][0]
    e = [v for k, v in tok.event_end.items() if k.name == "TIME_SHIFT"][0]
    return s, e
