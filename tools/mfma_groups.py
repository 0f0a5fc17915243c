#!/usr/bin/env python3
"""Time k_fwd_mfma one work group (octave) at a time: 10 traces x 131072 (debug knob TSPWS_MFMA_ONLY_GROUP)."""
import os, subprocess, sys, json
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import importlib, os, sys, time
sys.path.insert(0, os.path.join(%r, ".."))
sys.path.insert(0, os.path.join(%r, "..", "tests"))
import numpy as np, torch, abi, ctypes as C
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
N, K = 131072, int(os.environ.get("MG_TRACES", "10"))
p = tspws.resolve(abi.default_params(Kmax=K), N)
pl = tspws.Plan(p, N)
x = torch.randn(K, N, dtype=torch.float64, device="cuda")
Y = torch.empty((K, 2 * pl.ncoef), dtype=torch.float64, device="cuda")
def f(): tspws.check(lib.tspws_hip_forward_f64(pl.h, x.data_ptr(), K, N, Y.data_ptr(), None))
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); print("%%.1f" %% ((time.perf_counter() - t0) / 10 * 1e6))
''' % (here, here)
groups = [int(a) for a in sys.argv[1:]] or ([-1] + list(range(14)))
for g in groups:
    env = dict(os.environ); env.setdefault("TSPWS_FWD_KERNEL", "mfma")
    if g >= 0: env["TSPWS_MFMA_ONLY_GROUP"] = str(g)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if os.environ.get("MG_VERBOSE"):
        print("\n".join(l for l in out.stdout.splitlines() if l.startswith("blk"))[:6000])
        print("\n".join(l for l in out.stderr.splitlines() if l.startswith("mfma"))[:6000])
    print("group", g, "us per forward incl. gather:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
