#!/usr/bin/env python3
"""Float outputs of one single-stage call with the engine pinned by TSPWS_ENGINE, saved for a comparison across processes; with `cmp` the two
saved runs and the oracle are compared.  usage: TSPWS_ENGINE=fir engine_diff.py run mtr N out.npz | engine_diff.py cmp a.npz b.npz mtr N"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, abi
if sys.argv[1] == "run":
    import torch
    tspws = importlib.import_module("ts-pws_amd")
    mtr, N = int(sys.argv[2]), int(sys.argv[3])
    pl = tspws.Plan(tspws.resolve(abi.default_params(), N), N)
    X = tspws.synth(mtr, N, seed=1)
    ls, ts = pl.stack_single(X)
    torch.cuda.synchronize()
    np.savez(sys.argv[4], ls=ls.cpu().numpy(), ts=ts.cpu().numpy(), X=X.cpu().numpy())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    mtr, N = int(sys.argv[4]), int(sys.argv[5])
    for k in ("ls", "ts"):
        d = np.nonzero(a[k] != b[k])[0]
        print(k, "differing samples", d.size, "relerr between engines", abi.relerr(a[k], b[k]), "first", d[:5], [ (float(a[k][i]), float(b[k][i])) for i in d[:3]])
    w = abi.run_main(abi.oracle().orc_tspws_main_mt, abi.default_params(), a["X"])
    for name, r in (("fir", a), ("spectral", b)):
        print(name, "vs oracle: ls", abi.relerr(r["ls"], w["ls"]), "tsPWS", abi.relerr(r["ts"], w["tsPWS"]), "bit-equal", np.array_equal(r["ls"], w["ls"]), np.array_equal(r["ts"], w["tsPWS"]))
