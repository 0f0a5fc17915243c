#!/usr/bin/env python3
"""Times random subsampling (single-stage, M masks) on HBM-resident traces.   usage: sub_bench.py [mtr] [N] [M]"""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hashlib
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 499
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16501
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
p = tspws.resolve(abi.default_params(subsmpl_N=M, subsmpl_p=0.5), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
jl = torch.empty((M, N), dtype=torch.float32, device="cuda")
jt = torch.empty((M, N), dtype=torch.float32, device="cuda")


def run():
    abi.srand(1)
    tspws.check(lib.tspws_hip_subsample(pl.h, C.byref(pl.params), X.data_ptr(), N, mtr, M, jl.data_ptr(), jt.data_ptr(), None), "subsample")


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    run()
torch.cuda.synchronize()
h = hashlib.sha1(jt.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"subsampling {mtr}x{N} M={M}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/call, digest {h}")
