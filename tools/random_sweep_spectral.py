#!/usr/bin/env python3
"""Seeded sweep of the spectral forward engine against the oracle: whole tspws_main calls on random frames (Morlet / exact Morlet /
Mexican hat, V = 2 .. 8, w0, b0, s0, J), N = 1024 .. 16384 (powers of two: frames with a spectral set; every eighth case another N),
1 .. 300 traces with zero traces / stretches of exact zeros / a DC offset, single-stage (wu, unbiased, rm), two-stage with many groups
(the FP64 partial stacks as a many-trace batch) and stack + jackknife calls with >= 64 rows (rows in columns).  The engine is pinned by
the environment of the process (TSPWS_ENGINE=spectral, TSPWS_SPEC_NSMAX / TSPWS_FEW_NSMAX from the -DTSPWS_SWEEPS build):
  usage: TSPWS_LIB_PATH=ts-pws_amd/lib/libtspws_hip_sweeps.so TSPWS_ENGINE=spectral [TSPWS_SPEC_NSMAX=n] random_sweep_spectral.py [first_seed [n_seeds]]
SWEEP_ANYN=1 (round 6): trace lengths that are not powers of two in every case (the window of the periodic extension, the dense contraction
for the scales that do not fit it; TSPWS_SPEC_NT=min|double, TSPWS_GEMM=0, TSPWS_GEMM_KS=n, TSPWS_GEMM_ORDER=0|1|2 on the sweeps build)."""
import importlib
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = n = 0
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    for it in range(4):
        kw = {}
        typ = int(rng.choice([-1, -1, -1, -2, -3]))
        if typ != -1:
            kw["type"] = typ
        if typ != -3 and rng.random() < 0.5:
            kw["w0"] = float(rng.uniform(4.0, 9.0))
        if rng.random() < 0.3:
            kw["V"] = int(rng.integers(2, 9))
        if rng.random() < 0.3:
            kw["b0"] = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
        if rng.random() < 0.2:
            kw["J"] = int(rng.integers(3, 9))
        if os.environ.get("SWEEP_ANYN"):   # round 6: ANY trace length -- odd, just above / below a power of two, 3 * 2^k, the shipped example's 16501
            N = int(rng.choice([int(rng.integers(1024, 20000)), int(2 ** rng.integers(10, 15)) + int(rng.integers(-3, 4)), 3 * int(2 ** rng.integers(9, 13)), 16501, 4097, 8191]))
            N = max(N, 1024)
        else:
            N = int(rng.choice([1024, 2048, 4096, 8192, 16384])) if (it + seed) % 8 else int(rng.choice([1500, 3000, 5000, 12288]))
        mode = int(rng.integers(0, 4))
        times = None
        if mode == 0:     # single-stage
            mtr = int(rng.integers(1, 300))
            kw.update(wu=float(rng.choice([2.0, 2.0, 1.0, 1.5])), unbiased=int(rng.random() < 0.3), lrm=int(rng.random() < 0.2))
        elif mode == 1:   # two-stage, many groups: the partial stacks are a many-trace batch of FP64 rows
            K = int(rng.integers(64, 200))
            mtr = int(rng.integers(K, 3 * K))
            kw.update(Kmax=K, unbiased=int(rng.random() < 0.5))
        elif mode == 2:   # two-stage, few groups (few-trace kernels; the engine switch must not matter)
            mtr = int(rng.integers(8, 120))
            kw.update(Kmax=int(rng.integers(2, 9)), unbiased=int(rng.random() < 0.5))
        else:             # stack + jackknife: (C + 1) Kmax rows in columns
            nb, d = (int(rng.integers(6, 11)), 1) if rng.random() < 0.7 else (int(rng.integers(4, 7)), 2)
            K = int(rng.integers(6, 14))
            mtr = int(rng.integers(40 * nb, 60 * nb))
            kw.update(Kmax=K, unbiased=int(rng.random() < 0.5), jackknife_n=nb, jackknife_d=d)
            times = (1262304000 + 86400 * np.sort(rng.integers(0, 365, mtr))).astype(np.int64)
            N = min(N, 8192)
        X = abi.synth_traces(mtr, N, seed=seed * 10 + it)
        if rng.random() < 0.5:
            X[int(rng.integers(0, mtr))] = 0
        if rng.random() < 0.5:
            a0 = int(rng.integers(0, N // 2)); X[int(rng.integers(0, mtr)), a0:a0 + int(rng.integers(1, N // 2))] = 0
        if rng.random() < 0.2:
            X += np.float32(rng.uniform(-3, 3))
        p = abi.default_params(**kw)
        a = abi.run_main(lib.tspws_main, p, X, times=times)
        b = abi.run_main(abi.oracle().orc_tspws_main, p, X, times=times)
        n += 1
        ok = a["rc"] == b["rc"] and all(getattr(a["params"], f) == getattr(b["params"], f) for f in ("J", "V", "fold", "s0", "b0", "w0"))
        if ok and a["rc"] == 0:
            ok = abi.relerr(a["ls"], b["ls"]) < 2e-6 and abi.relerr(a["tsPWS"], b["tsPWS"]) < 2e-6
            if ok and times is not None:
                ok = np.array_equal(a["jk_mtr"], b["jk_mtr"]) and all(abi.relerr(a["jk_ts"][c], b["jk_ts"][c]) < 2e-6 and abi.relerr(a["jk_ls"][c], b["jk_ls"][c]) < 2e-6
                                                                         for c in range(len(a["jk_mtr"])))
        if not ok:
            bad += 1
            print("MISMATCH", seed, it, kw, N, mtr, a["rc"], b["rc"], flush=True)
print("cases", n, "mismatches", bad, "anyN", os.environ.get("SWEEP_ANYN"), "NT", os.environ.get("TSPWS_SPEC_NT"), "gemm", os.environ.get("TSPWS_GEMM"), os.environ.get("TSPWS_GEMM_KS"), os.environ.get("TSPWS_GEMM_ORDER"), "engine", os.environ.get("TSPWS_ENGINE"), "nsmax", os.environ.get("TSPWS_SPEC_NSMAX"), os.environ.get("TSPWS_FEW_NSMAX"), "few min", os.environ.get("TSPWS_FEW_SPEC_MIN"))
