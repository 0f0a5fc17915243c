#!/usr/bin/env python3
"""Seeded sweep over the optional outputs of tspws_main -- convergence curves (with / without AllSteps, own reference trace),
random subsampling, two-stage jackknife, Nmax -- on top of random frame parameters: this engine against the oracle.
usage: random_sweep_features.py [first_seed [n_seeds]]"""
import importlib, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, abi
import test_hip_parity as T
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40
TOL = 2e-6
bad = n = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(5000 + seed)
    for it in range(4):
        kw, N, mtr, beg = T._random_case(rng)
        N = int(rng.choice([256, 509, 1000, 1024, 2048, 3001])); mtr = int(rng.choice([3, 8, 17, 40]))
        kw.pop("fold", None); beg = 0.0
        feat = int(rng.integers(0, 4))
        times = ref_tr = None
        if feat == 0:
            kw["convergence"] = 1
            if rng.random() < 0.5: kw["AllSteps"] = 1
            if rng.random() < 0.4: ref_tr = abi.synth_traces(1, N, seed=999 + seed)[0]
        elif feat == 1:
            kw["subsmpl_N"] = int(rng.integers(1, 5)); kw["subsmpl_p"] = float(rng.choice([0.2, 0.5, 0.8, 1.0]))
        elif feat == 2:
            nb = int(rng.integers(3, 7)); d = int(rng.integers(1, min(3, nb)))
            kw.update(jackknife_n=nb, jackknife_d=d); kw.setdefault("Kmax", int(rng.integers(1, 6)))
            times = 1262304000 + 86400 * np.sort(rng.integers(0, 2 * 365, mtr))
            if rng.random() < 0.3: times = rng.permutation(times)
        else:
            kw["Nmax"] = int(rng.integers(1, mtr + 3))     # may exceed the trace count: clamped (the reference reads past the end)
        X = abi.synth_traces(mtr, N, seed=77 * seed + it)
        p = abi.default_params(**kw)
        if kw.get("Nmax", 0) > mtr:
            pb = abi.default_params(**dict(kw, Nmax=mtr))   # the oracle is asked for what the engine clamps to
        else:
            pb = p
        abi.srand(3); a = abi.run_main(lib.tspws_main, p, X, beg=beg, times=times, reference=ref_tr)
        abi.srand(3); b = abi.run_main(abi.oracle().orc_tspws_main, pb, X, beg=beg, times=times, reference=ref_tr)
        n += 1
        msgs = []
        if a["rc"] != b["rc"]: msgs.append(f"rc {a['rc']} vs {b['rc']}")
        elif a["rc"] == 0:
            if abi.relerr(a["ls"], b["ls"]) >= TOL or abi.relerr(a["tsPWS"], b["tsPWS"]) >= TOL: msgs.append("main outputs")
            for key in ("jk", "sub"):
                if key + "_ls" in a:
                    if key == "jk" and not np.array_equal(a["jk_mtr"], b["jk_mtr"]): msgs.append("jk_mtr")
                    for m in range(a[key + "_ls"].shape[0]):
                        if key == "jk" and not b["jk_mtr"][m]: continue
                        sc_l = np.max(np.abs(b[key + "_ls"][m])); sc_t = np.max(np.abs(b[key + "_ts"][m]))
                        if sc_l and abi.relerr(a[key + "_ls"][m], b[key + "_ls"][m]) >= TOL: msgs.append(f"{key}_ls[{m}]")
                        if sc_t and abi.relerr(a[key + "_ts"][m], b[key + "_ts"][m]) >= TOL: msgs.append(f"{key}_ts[{m}]")
            if "conv_ls_sim" in a:
                for k in ("conv_ls_sim", "conv_tsPWS_sim"):
                    if not np.allclose(a[k], b[k], atol=1e-8, rtol=0, equal_nan=True): msgs.append(k)
                for k in ("conv_ls_misfit", "conv_tsPWS_misfit"):
                    if not np.allclose(a[k], b[k], atol=1e-7 * np.nanmax(np.abs(b[k])) + 1e-18, rtol=0, equal_nan=True): msgs.append(k)
                if "conv_ts_steps" in a and (abi.relerr(a["conv_ts_steps"], b["conv_ts_steps"]) >= TOL or abi.relerr(a["conv_ls_steps"], b["conv_ls_steps"]) >= TOL):
                    msgs.append("steps")
        if msgs:
            bad += 1
            print("MISMATCH", seed, it, kw, "N", N, "mtr", mtr, msgs, flush=True)
print("cases", n, "mismatches", bad)
