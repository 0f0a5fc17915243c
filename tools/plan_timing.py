#!/usr/bin/env python3
"""Plan creation time (frame geometry + tap generation + tables), first and later plans of a process."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd")
tspws.load()
torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=10, unbiased=1), 131072), 131072)
    torch.cuda.synchronize()
    print(f"plan {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    pl.close()
