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


SWEEPS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ts-pws_amd", "lib", "libtspws_hip_sweeps.so")


@pytest.fixture(scope="session")
def sweeps():
    """(module, lib): a second instance of the binding over lib/libtspws_hip_sweeps.so -- the build (`make sweeps`, -DTSPWS_SWEEPS) whose
    tuning / A-B / test switches (TSPWS_TL_MIN, TSPWS_TL_BATCH, ...) are compiled in.  The shipped libtspws_hip.so does not read them, so
    tests that force a path by such a switch run on this one; child processes get it through TSPWS_LIB_PATH."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(SWEEPS_LIB), "build it: make -C ts-pws_amd sweeps"
    old = os.environ.get("TSPWS_LIB_PATH")
    os.environ["TSPWS_LIB_PATH"] = SWEEPS_LIB
    try:
        spec = importlib.util.spec_from_file_location("tspws_sweeps", os.path.join(root, "ts-pws_amd", "__init__.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if old is None:
            del os.environ["TSPWS_LIB_PATH"]
        else:
            os.environ["TSPWS_LIB_PATH"] = old
    return mod, mod.load()
