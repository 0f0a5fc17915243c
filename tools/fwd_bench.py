#!/usr/bin/env python3
"""Times the finish stage of the north-star call (the K forward transforms + phase stacks, weights, two inverses) on
HBM-resident partial stacks, and its forward part alone (tspws_hip_stacks_double).   usage: fwd_bench.py [N] [K] [reps]"""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
mtr = 100 * K
kw = dict(Kmax=K, unbiased=1)
if os.environ.get("FWD_MEXHAT"):
    kw["type"] = -3
p = tspws.resolve(abi.default_params(**kw), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda")
ts = torch.empty(N, dtype=torch.float32, device="cuda")
pl.stack_local(X, 0, mtr)
P = pl.reduce_buffer(mtr)
ST = torch.empty(2 * pl.ncoef, dtype=torch.float64, device="cuda")
PS = torch.empty(2 * pl.ncoef, dtype=torch.float64, device="cuda")


if os.environ.get("TSPWS_FWD_CLASSES"):  # -DFL_ABLATE=1 builds: switch classes of k_fwd_lds off (results wrong, only the clock counts)
    lib.tspws_hip_fwd_ablate()


if os.environ.get("TSPWS_INV_CLASSES"):  # likewise for the octave classes of k_inv_poly
    lib.tspws_hip_inv_ablate()


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


t_fin = timeit(lambda: pl.stack_finish(mtr, ls, ts))
t_fwd = timeit(lambda: tspws.check(lib.tspws_hip_stacks_double(pl.h, P.data_ptr(), K, N, ST.data_ptr(), PS.data_ptr(), None), "stacks_double"))
import hashlib
pl.stack_finish(mtr, ls, ts)
torch.cuda.synchronize()
h = hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"N={N} K={K} {'mexhat' if 'type' in kw else 'morlet'}: finish stage {t_fin:.1f} us, forward + stacks {t_fwd:.1f} us, digest {h}")
