import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:                      # tests/rt_diag.py (shared with tools/roundtrip_stress.py)
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle
