#!/usr/bin/env python3
"""What the three HIP events of a profiled call (bench.py: start, end of the streaming stage, end) cost the call: wall clock per
call over 40 calls with and without them.  usage: prof_cost.py"""
import sys, os, time, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
N, mtr = 131072, 10000
p = tspws.resolve(abi.default_params(Kmax=10, unbiased=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): pl.stack_single(X, ls, ts)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(5)
for rep in range(3):
    a = run(40)
    pl.profile_begin(40); b = run(40); st, call = pl.profile_read()
    print(f"without events {a:.4f} ms/call, with the three profile events {b:.4f} ms/call (GPU per call median {sorted(call)[len(call)//2]:.4f})")
