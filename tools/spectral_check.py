#!/usr/bin/env python3
"""Spectral forward engine (csrc/spectral.hip) against the oracle: per-trace coefficients of the spectral scales for float and
double input on several frames, then whole single-stage calls with the engine pinned.   usage: spectral_check.py [quick]"""
import importlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()


def dev_forward_spectral(pl, X, nsmax):
    ntr, N = X.shape
    Xd = torch.as_tensor(X, device="cuda")
    Y = torch.zeros((ntr, 2 * pl.ncoef), dtype=torch.float64, device="cuda")
    fn = lib.tspws_hip_forward_spectral_f64 if X.dtype == np.float64 else lib.tspws_hip_forward_spectral_f32
    tspws.check(fn(pl.h, Xd.data_ptr(), ntr, N, Y.data_ptr(), nsmax, None), "forward_spectral")
    torch.cuda.synchronize()
    return Y.cpu().numpy().view(np.complex128)


def coefficients():
    worst = 0
    cases = [(dict(), 1024, 3, 1 << 20), (dict(), 4096, 70, 256), (dict(), 131072, 2, 1 << 20), (dict(type=-3), 32768, 5, 4096), (dict(w0=2 * np.pi), 32768, 7, 1024),
             (dict(type=-2), 8192, 2, 1 << 20), (dict(b0=4.0), 65536, 2, 2048), (dict(J=3), 2048, 4, 1 << 20), (dict(V=7), 4096, 130, 1 << 20), (dict(), 1 << 20, 1, 1 << 20)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cases = cases[:5]
    for kw, N, ntr, nsmax in cases:
        p = abi.resolve(abi.default_params(**kw), N)
        f = abi.OracleFrame.from_params(p, N)
        pl = tspws.Plan(p, N)
        sf = lib.tspws_hip_spectral_first_scale(pl.h, nsmax)
        X = abi.synth_traces(ntr, N, seed=5)
        X[ntr // 2, N // 3: N // 3 + N // 4] = 0    # a stretch of exact zeros
        Y = dev_forward_spectral(pl, X.astype(np.float64), nsmax)
        Y32 = dev_forward_spectral(pl, X, nsmax)
        off = np.concatenate([[0], np.cumsum(f.Ns.astype(np.int64))])
        e = 0
        for t in sorted(set([0, ntr // 2, ntr - 1])):
            Yo = f.forward(X[t].astype(np.float64))
            for s in range(sf, f.S):
                a, b = int(off[s]), int(off[s + 1])
                es = max(abi.relerr(Y[t][a:b], Yo[a:b]), abi.relerr(Y32[t][a:b], Yo[a:b]))
                e = max(e, es)
                if es > 1e-11:
                    print("   trace", t, "scale", s, "D", f.D[s], "Ns", f.Ns[s], "L", f.L[s], "relerr", es)
        print(f"{str(kw):28s} N={N:8d} ntr={ntr:4d} nsmax={nsmax:8d}: spectral scales [{sf}, {f.S}) coefficients {e:.2e}", flush=True)
        worst = max(worst, e)
        assert sf < f.S and e < 1e-11
    print("worst", worst)


def whole_calls():
    """single-stage calls in fresh processes with the engine pinned, against the oracle"""
    code = r'''
import importlib, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, abi
tspws = importlib.import_module("ts-pws_amd"); tspws.load()
for kw, N, ntr in [(dict(), 4096, 100), (dict(w0=2 * np.pi), 8192, 200), (dict(type=-3), 4096, 70), (dict(unbiased=1), 2048, 300), (dict(wu=1.5), 1024, 64)]:
    pk = abi.default_params(**kw)
    X = abi.synth_traces(ntr, N, seed=9)
    X[3] = 0
    X[5, N // 4: N // 2] = 0
    pl = tspws.Plan(tspws.resolve(pk, N), N)
    ls, ts = pl.stack(torch.as_tensor(X, device="cuda"))
    torch.cuda.synchronize()
    want = abi.run_main(abi.oracle().orc_tspws_main, pk, X)
    e = max(abi.relerr(ls.cpu().numpy(), want["ls"]), abi.relerr(ts.cpu().numpy(), want["tsPWS"]))
    print(f"   {os.environ.get('TSPWS_ENGINE')}: {str(kw):24s} N={N} ntr={ntr}: whole call {e:.2e}", flush=True)
    assert e < 2e-6
''' % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    for eng in ("spectral", "fir"):
        env = dict(os.environ, TSPWS_ENGINE=eng)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout + r.stderr[-2000:])
        assert r.returncode == 0


coefficients()
whole_calls()
