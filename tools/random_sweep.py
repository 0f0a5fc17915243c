#!/usr/bin/env python3
"""Extended seeded sweep over the parameter grammar (tests/test_hip_parity.py::_random_case, 360 cases): tspws_main of this
engine against the oracle -- return codes, resolved parameters, mutated traces, both outputs; every third case with the many-trace
forward path forced (TSPWS_TL_MIN).  usage: random_sweep.py [first_seed [n_seeds]]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, importlib, abi
import test_hip_parity as T
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
bad = 0; n = 0
_a = int(sys.argv[1]) if len(sys.argv) > 1 else 8
_n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for seed in range(_a, _a + _n):
    rng = np.random.default_rng(1000 + seed)
    for it in range(6):
        kw, N, mtr, beg = T._random_case(rng)
        X = abi.synth_traces(mtr, N, seed=100 * seed + it)
        p = abi.default_params(**kw)
        # every third case through the many-trace forward path whatever its size (the path choice is read at every call)
        if it % 3 == 2:
            os.environ["TSPWS_TL_MIN"] = "2"
        else:
            os.environ.pop("TSPWS_TL_MIN", None)
        a = abi.run_main(lib.tspws_main, p, X, beg=beg)
        b = abi.run_main(abi.oracle().orc_tspws_main, p, X, beg=beg)
        n += 1
        ok = a["rc"] == b["rc"] and all(getattr(a["params"], f) == getattr(b["params"], f) for f in ("J", "V", "fold", "s0", "b0", "w0"))
        ok = ok and np.array_equal(a["sigall"], b["sigall"])
        if ok and a["rc"] == 0:
            ok = abi.relerr(a["ls"], b["ls"]) < 2e-6 and abi.relerr(a["tsPWS"], b["tsPWS"]) < 2e-6
        if not ok:
            bad += 1
            print("MISMATCH", seed, it, kw, N, mtr, beg, a["rc"], b["rc"], flush=True)
print("cases", n, "mismatches", bad)
