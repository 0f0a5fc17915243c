import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return {k: np.load(os.path.join(here, k + ".npz"), allow_pickle=False) for k in ("frames", "cwt", "mains", "example32", "extra")}
