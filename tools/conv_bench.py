#!/usr/bin/env python3
"""Times the convergence curves (one ts-PWS per prefix of the ensemble) through tspws_main on host traces.   usage: conv_bench.py [mtr] [N]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hashlib
import numpy as np
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 499
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16501
X = abi.synth_traces(mtr, N, seed=3)
p = abi.default_params(convergence=1)
abi.run_main(lib.tspws_main, p, X)
t0 = time.perf_counter()
r = abi.run_main(lib.tspws_main, abi.default_params(convergence=1), X)
dt = time.perf_counter() - t0
t1 = time.perf_counter()
abi.run_main(lib.tspws_main, abi.default_params(), X)
d0 = time.perf_counter() - t1
h = hashlib.sha1(np.ascontiguousarray(r["conv_tsPWS_sim"]).tobytes() + np.ascontiguousarray(r["conv_tsPWS_misfit"]).tobytes()).hexdigest()[:12]
print(f"convergence {mtr}x{N}: {(dt - d0) * 1e3:.1f} ms for the {mtr} prefix stacks (whole call {dt * 1e3:.1f} ms, without curves {d0 * 1e3:.1f} ms), digest {h}")
