#!/usr/bin/env python3
"""k_rows_walk alone (Plan.jackknife_local = the one-pass rows of cfg4's stack + 10 replicas, nothing else on the device) against the same
walk inside the whole stack + jackknife call: is the 0.86 ms the kernel or the state of the chip behind 1.4 ms of FP64 kernels?"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
N, mtr, Cn = 131072, 10000, 10
p = tspws.resolve(abi.default_params(type=-3, Kmax=10, jackknife_n=10, jackknife_d=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)
sel = np.zeros((Cn, mtr), np.int8)
assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, 10, Cn) == 0
pl.jackknife_buffer(Cn)
for rep in range(3):
    for _ in range(3):
        pl.jackknife_local(X, 0, mtr, sel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pl.jackknife_local(X, 0, mtr, sel)
    torch.cuda.synchronize()
    print("rows of the stack + 10 replicas alone: %.1f us per call (walk + fix-up + a 10-MB copy, synchronous)" % ((time.perf_counter() - t0) / 20 * 1e6))
    time.sleep(1.0)
