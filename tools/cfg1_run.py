#!/usr/bin/env python3
"""BASELINE configs[0]-shaped run: 499 traces x 16501 samples (the shipped example's shape; N is odd, so no decimation
divides it), default Morlet single-stage ts-PWS and the TwoStage=10 unbiased variant; for rocprofv3."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
tspws.load()
N, mtr = 16501, 499
X = tspws.synth(mtr, N, seed=1)
for name, kw in (("single-stage", dict()), ("two-stage K=10 unbiased", dict(Kmax=10, unbiased=1))):
    p = tspws.resolve(abi.default_params(**kw), N)
    pl = tspws.Plan(p, N)
    for _ in range(2):
        pl.stack(X)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pl.stack(X)
    torch.cuda.synchronize()
    print(f"cfg1 {name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/call  (V={p.V} J={p.J})")
