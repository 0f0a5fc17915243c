#!/usr/bin/env python3
"""End-to-end time of the drop-in tspws_main on HOST buffers (upload over PCIe included), next to the
reference CPU library when oracle/_ref is present.   usage: host_path_timing.py [mtr] [N]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
X = np.random.default_rng(0).uniform(-0.5, 0.5, (mtr, N)).astype(np.float32)
p = abi.default_params(Kmax=10, unbiased=1)
import ctypes as C


def call(fn, X):
    """time only the C call (no numpy copies inside the timed region)"""
    p_ = abi.t_tsPWS.from_buffer_copy(p)
    out = abi.t_tsPWS_out()
    ls = np.zeros(N, np.float32)
    ts = np.zeros(N, np.float32)
    out.ls = ls.ctypes.data_as(C.POINTER(C.c_float))
    out.tsPWS = ts.ctypes.data_as(C.POINTER(C.c_float))
    d = abi.t_data()
    d.sigall = X.ctypes.data_as(C.POINTER(C.c_float))
    d.hdr.max, d.hdr.mtr, d.hdr.dt, d.hdr.beg = N, mtr, 1.0, 0.0
    t0 = time.perf_counter()
    rc = fn(C.byref(p_), C.byref(out), C.byref(d))
    return time.perf_counter() - t0, rc, ts


res = {}
for name, fn in (("hip tspws_main (host buffers)", lib.tspws_main), ("reference", abi.ref().tspws_main if abi.ref() else None)):
    if fn is None:
        continue
    for rep in range(3):
        dt, rc, ts = call(fn, X)
    res[name] = ts
    print(f"{name:32s} {mtr} x {N}: {dt * 1e3:9.3f} ms  ({mtr * N / dt:.3e} samples/s, {4 * mtr * N / dt / 1e9:.1f} GB/s of input)  rc={rc}")
# a caller that hands a NEW buffer every call (fresh pages, never pinned before)
for rep in range(3):
    Y = X.copy()
    dt, rc, ts = call(lib.tspws_main, Y)
    del Y
print(f"{'hip tspws_main (fresh buffer)':32s} {mtr} x {N}: {dt * 1e3:9.3f} ms  ({4 * mtr * N / dt / 1e9:.1f} GB/s of input)  rc={rc}")
if len(res) == 2:
    a, b = res.values()
    print("relerr hip vs reference:", abi.relerr(a, b))
