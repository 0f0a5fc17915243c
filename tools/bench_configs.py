#!/usr/bin/env python3
"""Extra timings for the other BASELINE.json configs (not the driver's bench contract):
cfg2 single-stage 1024x32768 (w0=2pi), cfg3 two-stage, cfg4 MexHat two-stage + jackknife n=10 d=1."""
import ctypes as C
import importlib
import json
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {}
# cfg2
N, mtr = 32768, 1024
p = tspws.resolve(abi.default_params(w0=2 * np.pi), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
t = timeit(lambda: pl.stack(X))
res["cfg2_single_stage_1024x32768_w2pi"] = dict(ms=t * 1e3, samples_per_s=mtr * N / t, V=p.V, J=p.J, S=pl.S)
del X, pl
# cfg4: jackknife
N, mtr = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = tspws.resolve(abi.default_params(type=-3, Kmax=10, jackknife_n=10, jackknife_d=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)
Cn = 10
sel = np.zeros((Cn, mtr), np.int8)
assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, 10, Cn) == 0


def jk():
    # the stack and its ten replicas from ONE pass over the traces
    pl.stack_jackknife(X, sel)



t = timeit(jk, n=3, warm=1)
res["cfg4_mexhat_twostage_jackknife_n10_d1"] = dict(ms=t * 1e3, samples_per_s=mtr * N / t, replicas=Cn, mtr=mtr)
print(json.dumps(res))
