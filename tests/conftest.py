import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    p = load_package()
    p.build()
    return p


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    import checkers
    checkers.build_oracle()


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name))
