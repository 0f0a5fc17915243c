#!/usr/bin/env python3
"""Times one whole call of a named configuration on HBM-resident traces and prints a digest of the float outputs.
usage: cfg_bench.py cfg1|cfg2|cfg3|cfg1s [reps]     (cfg1 = the shipped example's shape 499 x 16501 single-stage)"""
import hashlib
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
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which.startswith("c:"):   # c:mtr:N[:mexhat]
    f = which.split(":")
    table = {which: (dict(type=-3) if len(f) > 3 else dict(), int(f[1]), int(f[2]))}
else:
    table = None
kw, mtr, N = table[which] if table else {"cfg1": (dict(), 499, 16501), "cfg2": (dict(w0=2 * np.pi), 1024, 32768), "cfg3": (dict(Kmax=10, unbiased=1), 10000, 131072),
              "cfg2d": (dict(), 1024, 32768), "cfg2m": (dict(type=-3), 1024, 32768)}[which]
p = tspws.resolve(abi.default_params(**kw), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda")
ts = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(2):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
h = hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{which} {mtr}x{N} V={p.V} J={p.J}: {dt * 1e3:.3f} ms/call, digest {h}")
