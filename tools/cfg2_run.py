#!/usr/bin/env python3
"""BASELINE configs[1]-like single-stage run (1024 x 32768, w0 = 2 pi) a few times; for rocprofv3."""
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
N, mtr = 32768, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = tspws.resolve(abi.default_params(w0=2 * np.pi), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
for _ in range(3):
    pl.stack(X)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pl.stack(X)
torch.cuda.synchronize()
print("cfg2 ms/call", (time.perf_counter() - t0) / 5 * 1e3)
