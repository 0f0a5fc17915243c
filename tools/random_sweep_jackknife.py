#!/usr/bin/env python3
"""Seeded sweep over the trace-sharded jackknife (tspws_hip_jackknife_local / _finish): random shards (empty ones included),
random deletion plans, replicas finished in random ranges -- against the oracle's tspws_main.
usage: random_sweep_jackknife.py [first_seed [n_seeds]]"""
import importlib, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, abi
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = n = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(7000 + seed)
    for it in range(3):
        N = int(rng.choice([1024, 2048, 3001, 4096])); mtr = int(rng.choice([30, 77, 150, 260]))
        nb = int(rng.integers(3, 8)); d = int(rng.integers(1, min(3, nb)))
        kw = dict(Kmax=int(rng.integers(1, 9)), jackknife_n=nb, jackknife_d=d, type=int(rng.choice([-1, -1, -3])), wu=float(rng.choice([2.0, 1.0, 1.5])))
        if kw["wu"] == 2.0 and rng.random() < 0.5: kw["unbiased"] = 1
        times = (1262304000 + 86400 * np.sort(rng.integers(0, 3 * 365, mtr))).astype(np.int64)
        if rng.random() < 0.3: times = rng.permutation(times)
        p = tspws.resolve(abi.default_params(**kw), N)
        K = p.Kmax
        if K > mtr: continue
        pl = tspws.Plan(p, N)
        X = tspws.synth(mtr, N, seed=seed * 7 + it)
        Cn = abi.binomial(nb, d)
        sel = np.zeros((Cn, mtr), np.int8)
        assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, d, nb, Cn) == 0
        cuts = sorted(set([0, mtr] + [int(c) for c in rng.integers(0, mtr + 1, int(rng.integers(1, 4)))]))
        if rng.random() < 0.3: cuts = sorted(cuts + [cuts[-2]])
        main = torch.zeros(K * N, dtype=torch.float64, device="cuda"); rows = torch.zeros(Cn * K * N, dtype=torch.float64, device="cuda")
        for a, b in zip(cuts[:-1], cuts[1:]):
            pl.jackknife_local(X[a:b], a, mtr, sel); torch.cuda.synchronize()
            main += pl.reduce_buffer(mtr); rows += pl.jackknife_buffer(Cn)
        pl.reduce_buffer(mtr).copy_(main); pl.jackknife_buffer(Cn).copy_(rows)
        ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
        pl.stack_finish(mtr, ls, ts)
        jl = torch.zeros((Cn, N), dtype=torch.float32, device="cuda"); jt = torch.zeros((Cn, N), dtype=torch.float32, device="cuda")
        jm = np.zeros(Cn, np.uint32)
        cut = int(rng.integers(0, Cn + 1))
        pl.jackknife_finish(mtr, sel, cut, Cn, jl, jt, jm); pl.jackknife_finish(mtr, sel, 0, cut, jl, jt, jm)
        torch.cuda.synchronize()
        want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X.cpu().numpy(), times=times)
        n += 1
        msgs = []
        if not np.array_equal(jm, want["jk_mtr"]): msgs.append("jk_mtr")
        if abi.relerr(ls.cpu().numpy(), want["ls"]) >= 2e-6 or abi.relerr(ts.cpu().numpy(), want["tsPWS"]) >= 2e-6: msgs.append("main")
        for c in range(Cn):
            if not want["jk_mtr"][c]: continue
            if abi.relerr(jl[c].cpu().numpy(), want["jk_ls"][c]) >= 2e-6: msgs.append(f"ls[{c}]")
            if np.max(np.abs(want["jk_ts"][c])) and abi.relerr(jt[c].cpu().numpy(), want["jk_ts"][c]) >= 2e-6: msgs.append(f"ts[{c}]")
        if msgs:
            bad += 1
            print("MISMATCH", seed, it, kw, "N", N, "mtr", mtr, "cuts", cuts, msgs, flush=True)
        pl.close()
print("jackknife cases", n, "mismatches", bad)
