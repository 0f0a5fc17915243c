#!/usr/bin/env python3
"""Headline call (10 000 x 131 072, two-stage K = 10, unbiased) with the library's own per-call events: median call, streaming stage and finish stage
(= call - streaming) over `reps` calls; for A/B runs of finish-stage variants in fresh processes.  usage: finish_ab.py [reps]"""
import importlib, os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, abi
tspws = importlib.import_module("ts-pws_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N, mtr = 131072, 10000
pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=10, unbiased=1), N), N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(5):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
pl.profile_begin(reps)
for _ in range(reps):
    pl.stack_single(X, ls, ts)
torch.cuda.synchronize()
stage, call = pl.profile_read()
h = hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"call {np.median(call)*1e3:.1f} us  streaming {np.median(stage)*1e3:.1f} us  finish {np.median(call - stage)*1e3:.1f} us  (min finish {np.min(call - stage)*1e3:.1f})  digest {h}  FEW_GEMM={os.environ.get('TSPWS_FEW_GEMM')} KS={os.environ.get('TSPWS_FEW_GEMM_KS')}")
