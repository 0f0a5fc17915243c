#!/usr/bin/env python3
"""Seeded sweep over the masked-replica engine (jackknife / two-stage random subsampling through tspws_main) at shapes the other
sweeps do not reach: many groups (Kmax up to 40), many replicas (C up to 56: the snapshot form), short and odd traces, trace counts
down to Kmax, unsorted and clustered start times, a reference trace for nothing -- this engine against the oracle.
usage: random_sweep_masked.py [first_seed [n_seeds]]"""
import importlib, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, abi
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40
TOL = 2e-6
bad = n = 0


def close(x, y):
    """relative error below TOL, non-finite values (a replica without traces divides by its zero trace count, like the reference) in the same places"""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    if not np.array_equal(np.isfinite(x), np.isfinite(y)):
        return False
    m = np.isfinite(y)
    return (not m.any()) or abi.relerr(np.where(m, x, 0.0), np.where(m, y, 0.0)) < TOL
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(7000 + seed)
    for it in range(3):
        N = int(rng.choice([64, 130, 509, 1000, 2048, 4096]))
        K = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40]))
        mtr = int(rng.integers(K, max(K + 1, 300)))
        kw = dict(Kmax=K, type=int(rng.choice([-1, -3])), unbiased=int(rng.integers(0, 2)), wu=float(rng.choice([2.0, 1.0, 1.5])))
        if rng.random() < 0.75:
            nb = int(rng.integers(2, 9)); d = int(rng.integers(1, min(4, nb)))
            kw.update(jackknife_n=nb, jackknife_d=d)
            span = int(rng.choice([30, 365, 3 * 365]))
            times = 1262304000 + 86400 * rng.integers(0, span, mtr)
            if rng.random() < 0.7: times = np.sort(times)
            if rng.random() < 0.2: times[: mtr // 2] = times[0]          # half the ensemble in one bin
        else:
            kw.update(subsmpl_N=int(rng.integers(1, 20)), subsmpl_p=float(rng.choice([0.1, 0.5, 0.9, 1.0])))
            times = None
        X = abi.synth_traces(mtr, N, seed=31 * seed + it)
        p = abi.default_params(**kw)
        abi.srand(seed)
        a = abi.run_main(lib.tspws_main, p, X, times=times)
        abi.srand(seed)
        b = abi.run_main(abi.oracle().orc_tspws_main, p, X, times=times)
        n += 1
        ok = a["rc"] == b["rc"]
        if ok and a["rc"] == 0:
            ok = close(a["ls"], b["ls"]) and close(a["tsPWS"], b["tsPWS"])
            for key in ("jk", "sub"):
                if key + "_ls" in a and ok:
                    if key == "jk": ok = bool(np.array_equal(a["jk_mtr"], b["jk_mtr"]))
                    for c in range(a[key + "_ls"].shape[0]):
                        ok = ok and close(a[key + "_ls"][c], b[key + "_ls"][c]) and close(a[key + "_ts"][c], b[key + "_ts"][c])
        if not ok:
            bad += 1
            print("MISMATCH", seed, it, kw, "N", N, "mtr", mtr, "rc", a["rc"], b["rc"])
print("masked cases", n, "mismatches", bad)
