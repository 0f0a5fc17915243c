#!/usr/bin/env python3
"""Streaming stage as the multi-GPU schedule issues it (two half-range launches) against the one-shot launch."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd")
N, mtr, K = 131072, 10000, 10
pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N), N)
X = tspws.synth(mtr, N, seed=1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one launch          %.3f ms" % timeit(lambda: pl.stack_local(X, 0, mtr)))
print("two half launches   %.3f ms" % timeit(lambda: (pl.partial_stacks_range(X, 0, mtr, 0, K // 2), pl.partial_stacks_range(X, 0, mtr, K // 2, K))))
print("five launches       %.3f ms" % timeit(lambda: [pl.partial_stacks_range(X, 0, mtr, g, g + 2) for g in range(0, K, 2)]))
