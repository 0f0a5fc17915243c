#!/usr/bin/env python3
"""tspws_main on very long traces (2^21 samples, an odd length above 2^20, Mexican hat) against the oracle."""
import importlib, sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
for kw, mtr, N in ((dict(Kmax=4, unbiased=1), 12, 1 << 21), (dict(), 3, (1 << 20) + 7), (dict(type=-3, Kmax=2), 6, 1 << 20)):
    X = abi.synth_traces(mtr, N, seed=5)
    p = abi.default_params(**kw)
    t0 = time.time(); a = abi.run_main(lib.tspws_main, p, X); t1 = time.time()
    b = abi.run_main(abi.oracle().orc_tspws_main_mt if hasattr(abi.oracle(), "orc_tspws_main_mt") else abi.oracle().orc_tspws_main, p, X); t2 = time.time()
    print(kw, mtr, N, "rc", a["rc"], b["rc"], "J", a["params"].J, "relerr ls %.2e ts %.2e" % (abi.relerr(a["ls"], b["ls"]), abi.relerr(a["tsPWS"], b["tsPWS"])), "gpu %.2fs cpu %.1fs" % (t1 - t0, t2 - t1), flush=True)
