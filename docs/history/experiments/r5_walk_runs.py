#!/usr/bin/env python3
"""k_rows_walk against the run structure of the selection (rocprofv3 timeline of this script): cfg4's jackknife selection (runs of ~30 traces:
the replicas' groups of 900 selected traces end elsewhere than the plain stack's groups of 1000) and all-ones replicas (every column has the
plain stack's groups: 10 runs of 1000 traces)."""
import importlib
import os
import sys

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
ones = np.ones((Cn, mtr), np.int8)
for s in (sel, ones, sel, ones, sel, ones):
    pl.stack_jackknife(X, s)
    torch.cuda.synchronize()
