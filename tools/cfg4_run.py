#!/usr/bin/env python3
"""BASELINE configs[3]: 10k x 131072, Mexican hat, two-stage K=10 + jackknife n=10 d=1; a few calls, for rocprofv3."""
import ctypes as C
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
lib = tspws.load()
N, mtr = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = tspws.resolve(abi.default_params(type=-3, Kmax=10, jackknife_n=10, jackknife_d=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)
Cn = 10
sel = np.zeros((Cn, mtr), np.int8)
assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, 10, Cn) == 0


if os.environ.get("TSPWS_FWD_CLASSES"):  # -DFL_ABLATE=1 builds: switch classes of k_fwd_lds off (results wrong, only the clock counts)
    lib.tspws_hip_fwd_ablate()


if os.environ.get("TSPWS_INV_CLASSES"):  # likewise for the octave classes of k_inv_poly
    lib.tspws_hip_inv_ablate()


def jk():
    # the stack and its ten replicas from ONE pass over the traces
    pl.stack_jackknife(X, sel)



jk()
torch.cuda.synchronize()
REPS = int(os.environ.get("CFG4_REPS", "3"))
t0 = time.perf_counter()
for _ in range(REPS):
    jk()
torch.cuda.synchronize()
same = (time.perf_counter() - t0) / REPS * 1e3
# a different selection every call (the host rebuilds its run lists and uploads them)
times_b = times + 86400 * 5
sel_b = np.zeros((Cn, mtr), np.int8)
assert lib.tspws_jackknife_plan(sel_b.ctypes.data, times_b.ctypes.data, mtr, 1, 10, Cn) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4):
    pl.stack_jackknife(X, sel_b if i % 2 == 0 else sel)
torch.cuda.synchronize()
print("cfg4 ms/call", same, "changed selection", (time.perf_counter() - t0) / 4 * 1e3)
