#!/usr/bin/env python3
"""One-off checks of the any-N spectral engine at sizes the test suite does not reach: several batches with the dense contraction (5000 x 16501), a long odd trace
(200 x 100003), the smallest admissible lengths (300 x 1024 / 1025 / 1100), against the multi-threaded oracle.  usage: big_anyN_check.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
for kw, mtr, N in ((dict(), 5000, 16501), (dict(), 200, 100003), (dict(wu=1.0), 300, 1024), (dict(), 300, 1025), (dict(type=-3), 300, 1100), (dict(Kmax=100, unbiased=1), 1000, 20001)):
    p = abi.default_params(**kw)
    pl = tspws.Plan(tspws.resolve(p, N), N)
    X = tspws.synth(mtr, N, seed=3)
    t0 = time.perf_counter()
    ls, ts = pl.stack(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = abi.run_main(abi.oracle().orc_tspws_main_mt, p, X.cpu().numpy())
    print(f"{kw} {mtr} x {N}: NT {lib.tspws_hip_spectral_transform_length(pl.h)} set [{lib.tspws_hip_spectral_choice(pl.h, mtr)}, {lib.tspws_hip_spectral_end_scale(pl.h)}) of {pl.S}  "
          f"relerr ls {abi.relerr(ls.cpu().numpy(), w['ls']):.2e} tsPWS {abi.relerr(ts.cpu().numpy(), w['tsPWS']):.2e}  first call {dt*1e3:.1f} ms", flush=True)
