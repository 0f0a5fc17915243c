#!/usr/bin/env python3
"""Per-phase wave timers of k_fwd_lds (debug build: tools/build_variant.sh tim "-DFL_TIMING=1", TSPWS_LIB_PATH=...).
usage: fwd_timing.py [cfg3|cfg2]   -- prints, per log2(D) class, shader-clock ticks per wave and trace in each phase."""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
lib.tspws_hip_fwd_timing.argtypes = [C.c_void_p, C.c_int]
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg2":
    N, mtr = 32768, 1024
    p = tspws.resolve(abi.default_params(w0=2 * np.pi), N)
elif which == "cfg3f":  # the ten transforms of cfg3, but straight from float traces (single-stage call on 10 traces)
    N, mtr = 131072, 10
    p = tspws.resolve(abi.default_params(), N)
else:
    N, mtr = 131072, 2000
    p = tspws.resolve(abi.default_params(Kmax=10, unbiased=1), N)
pl = tspws.Plan(p, N)
X = tspws.synth(mtr, N, seed=1)
pl.stack(X)
buf = np.zeros(56, np.uint64)
lib.tspws_hip_fwd_timing(buf.ctypes.data, 1)
pl.stack(X)
lib.tspws_hip_fwd_timing(buf.ctypes.data, 1)
b = buf.reshape(7, 8).astype(np.float64)
names = ["setup", "barrier1", "stage", "barrier2", "fma", "reduce"]
print(f"{which}: ticks per wave and trace (set-up: per wave)")
for c in range(7):
    if b[c, 7] == 0:
        continue
    tr, wv = b[c, 6], b[c, 7]
    row = "  ".join(f"{n} {b[c, i] / (wv if i == 0 else tr):8.0f}" for i, n in enumerate(names))
    tot = b[c, :6].sum() / tr
    print(f"logD {c}: waves {int(wv):6d} traces/wave {tr / wv:5.1f}  {row}  total/trace {tot:8.0f}  fma share {b[c, 4] / b[c, :6].sum():.2f}")
