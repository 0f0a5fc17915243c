#!/usr/bin/env python3
"""cfg4 call by call (30 calls), alone in the process or after five headline calls of ANOTHER plan (OTHER_PLANS=1): do the streams of the side-by-side transforms still run side by side?"""
import importlib, os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
N, mtr, Cn = 131072, 10000, 10
X = tspws.synth(mtr, N, seed=1)
if os.environ.get("OTHER_PLANS"):
    p0 = tspws.resolve(abi.default_params(Kmax=10, unbiased=1), N); pl0 = tspws.Plan(p0, N)
    for _ in range(5): pl0.stack_single(X)
    torch.cuda.synchronize()
p = tspws.resolve(abi.default_params(type=-3, Kmax=10, jackknife_n=10, jackknife_d=1), N)
pl = tspws.Plan(p, N)
times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)
sel = np.zeros((Cn, mtr), np.int8)
assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, 10, Cn) == 0
ts = []
for i in range(int(os.environ.get("NCALLS", "30"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl.stack_jackknife(X, sel)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
a = np.array(ts[1:]); print("calls %d  median %.3f  p10 %.3f  p90 %.3f  max %.3f | first 12:" % (len(a), np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.max()), " ".join("%.2f" % t for t in ts[1:13]), "| last 12:", " ".join("%.2f" % t for t in ts[-12:]))
