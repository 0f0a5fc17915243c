#!/usr/bin/env python3
"""What does an HBM-streaming kernel cost the forward transforms (and vice versa) when they share the GPU?
100 Mexican-hat transforms of 131072 samples (tspws_hip_stacks_double on 100 rows) beside torch's row sum over 2.6 GB of traces."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
N, K = 131072, 100
p = tspws.resolve(abi.default_params(type=-3, Kmax=K, unbiased=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(5000, N, seed=1)
P = torch.randn(K, N, dtype=torch.float64, device="cuda")
ST = torch.empty(2 * pl.ncoef, dtype=torch.float64, device="cuda")
PS = torch.empty(2 * pl.ncoef, dtype=torch.float64, device="cuda")
s2 = torch.cuda.Stream()
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


def fwd():
    tspws.check(lib.tspws_hip_stacks_double(pl.h, P.data_ptr(), K, N, ST.data_ptr(), PS.data_ptr(), None), "stacks_double")


def rd():
    return X.sum(dim=0)


for mode in ("fwd alone", "read alone", "both"):
    for rep in range(4):
        torch.cuda.synchronize()
        if mode != "read alone":
            e[0].record(); fwd(); e[1].record()
        if mode != "fwd alone":
            with torch.cuda.stream(s2):
                e[2].record(s2); rd(); rd(); e[3].record(s2)
        torch.cuda.synchronize()
        a = e[0].elapsed_time(e[1]) if mode != "read alone" else 0
        b = e[2].elapsed_time(e[3]) if mode != "fwd alone" else 0
        if rep:
            print(f"{mode:11s} fwd {a:.3f} ms   read {b:.3f} ms ({2 * X.numel() * 4 / (b * 1e-3) / 1e12 if b else 0:.2f} TB/s)")
