#!/usr/bin/env python3
"""Finish stage as two chains of scales on two streams (same kernels on sub-ranges of their launch lists) vs the plain one."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd")
N, mtr, K = 131072, 10000, 10
pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N), N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
xa = torch.empty(2 * N, dtype=torch.float64, device="cuda"); xb = torch.empty(2 * N, dtype=torch.float64, device="cuda")
pl.stack_local(X, 0, mtr)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("plain finish  %.4f ms" % timeit(lambda: pl.stack_finish(mtr, ls, ts)))
def two(cut):
    e = torch.cuda.Event()
    with torch.cuda.stream(s1):
        pl.stack_finish_scales(mtr, 0, cut, xa)
    with torch.cuda.stream(s2):
        pl.stack_finish_scales(mtr, cut, pl.S, xb)
        e.record(s2)
    s1.wait_event(e)
    with torch.cuda.stream(s1):
        xa.add_(xb); pl.epilogue(xa, mtr, ls, ts)
for cut in (8, 16, 20, 28, 36, 44):
    print("two chains, cut at scale %2d: %.4f ms" % (cut, timeit(lambda: two(cut))))
