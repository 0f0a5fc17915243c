"""Child process of test_short_frame_and_many_trace_forms_agree: whole tspws_main calls on short frames, small ensembles and a
many-trace batch against the oracle under the environment switches of the parent (read once per process / per frame by the
library).  Prints SHAPE_ENGINES <worst relative error> <digest of all float outputs>."""
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
worst = 0.0
h = hashlib.sha1()


def check(kw, mtr, N, seed):
    global worst
    X = abi.synth_traces(mtr, N, seed=seed)
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["rc"] == 0 and b["rc"] == 0, (a["rc"], b["rc"])
    worst = max(worst, abi.relerr(a["ls"], b["ls"]), abi.relerr(a["tsPWS"], b["tsPWS"]))
    h.update(a["ls"].tobytes()); h.update(a["tsPWS"].tobytes())


check(dict(), 70, 4096, 1)                              # single-stage, fused slices of a short frame
check(dict(), 33, 1501, 2)                              # N odd: three-frame inverse, one item per scale
check(dict(Kmax=10, unbiased=1), 64, 8192, 3)           # two-stage: ten partial stacks, one slice
check(dict(Kmax=4), 19, 3001, 4)
check(dict(type=-3, Kmax=5, unbiased=1), 40, 2048, 5)   # Mexican hat
check(dict(w0=2 * np.pi), 45, 16384, 6)                 # V = 5
check(dict(wu=1.5), 200, 2048, 7)                       # many traces of a short frame (few-trace kernels, several slices)
os.environ["TSPWS_TL_MIN"] = "64"                       # (read at every call) the many-trace path on a batch the rule would not give it
check(dict(), 130, 4096, 8)
check(dict(w0=2 * np.pi, unbiased=1), 70, 8192, 9)
print("SHAPE_ENGINES", worst, h.hexdigest()[:16])
