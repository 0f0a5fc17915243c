#!/usr/bin/env python3
"""A small ensemble (default 64 x 8192), single-stage and two-stage, a few calls each (for rocprofv3 timelines).  usage: small_run.py [mtr] [N]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd"); tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
for name, kw in (("two-stage K=10 unbiased", dict(Kmax=10, unbiased=1)), ("single-stage", dict())):
    p = tspws.resolve(abi.default_params(**kw), N)
    pl = tspws.Plan(p, N)
    for _ in range(3):
        pl.stack_single(X, ls, ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pl.stack_single(X, ls, ts)
    torch.cuda.synchronize()
    print(f"{mtr} x {N} {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/call")
