"""Run by test_hip_parity.py in a FRESH process: srand(), then the process's first tspws_main call asks for random subsamples.
The masks must be the ones the oracle draws from the same seed -- the HIP runtime initialises inside this very call and
consumes libc rand() values on the way (tspws_main draws its masks before the first HIP call)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
kw = dict(type=-3, wu=1.5, subsmpl_N=4, subsmpl_p=0.2)
N, mtr = 256, 17
X = abi.synth_traces(mtr, N, seed=5)
abi.srand(3)
a = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
abi.srand(3)
b = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X)
worst = max(max(abi.relerr(a["sub_ls"][m], b["sub_ls"][m]), abi.relerr(a["sub_ts"][m], b["sub_ts"][m])) for m in range(4))
print("FRESH_SUBSAMPLE", a["rc"], b["rc"], worst)
