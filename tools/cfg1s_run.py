#!/usr/bin/env python3
"""499 x 16501 single-stage only, a few calls (for rocprofv3 timelines)."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd"); tspws.load()
N, mtr = 16501, 499
X = tspws.synth(mtr, N, seed=1)
p = tspws.resolve(abi.default_params(), N)
pl = tspws.Plan(p, N)
for _ in range(6):
    pl.stack(X)
torch.cuda.synchronize()
