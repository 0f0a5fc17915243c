#!/usr/bin/env python3
"""One process = one build / one setting of the environment: times the north-star call and prints a digest of both float
outputs (and of two calls on CHANGED traces: a schedule that read stale partial stacks would show), so that variants can be
compared across processes.  Used for the round-3 overlap experiments (profiles/r03_overlap_experiments.txt).
usage: [TSPWS_LIB_PATH=... ] overlap_check.py [mtr] [N] [K] [steps]"""
import hashlib
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ctypes as C
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda")
ts = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(5):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
tspws.check(lib.tspws_hip_profile_begin(pl.h, steps), "profile_begin")
t0 = time.perf_counter()
for _ in range(steps):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ms, nc = C.c_double(), C.c_size_t()
tspws.check(lib.tspws_hip_profile_end(pl.h, C.byref(ms), C.byref(nc)), "profile_end")
h = hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:16]
# stale-data check: the traces change between calls (a schedule that let the transforms read the previous call's partial
# stacks would go unnoticed with constant input)
for _ in range(2):
    X[: mtr // 2] *= 0.5
    X[mtr // 2:] *= -1.0
    pl.stack_single(X, ls, ts)
    torch.cuda.synchronize()
    h += "/" + hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:8]
env = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TSPWS_"))
print(f"{env or 'default':60s} {mtr}x{N} K={K}: {dt * 1e3:.4f} ms/call, streaming stage {ms.value:.4f} ms, digest {h}")
