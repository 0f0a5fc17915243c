#!/usr/bin/env python3
"""A few LARGE random cases (16501 ... 131072 samples, 64 ... 500 traces, single- and two-stage, random frames) through the
device-resident entry against the oracle's parallel restatement.  usage: random_sweep_large.py [first_seed [n_seeds]]"""
import importlib, os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, abi
import test_hip_parity as T
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
orc = abi.oracle()
main = getattr(orc, "orc_tspws_main_mt", None) or orc.orc_tspws_main
main.restype = abi.C.c_int
bad = n = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(11000 + seed)
    kw, _, _, _ = T._random_case(rng)
    for k in ("fold", "lrm", "fmin"): kw.pop(k, None)
    N = int(rng.choice([16501, 32768, 50000, 65536, 131072])); mtr = int(rng.choice([64, 100, 128, 300, 500]))
    if N * mtr > 131072 * 130: mtr = 64
    if rng.random() < 0.5: kw["Kmax"] = int(rng.choice([4, 10, 16, 70]))
    p = tspws.resolve(abi.default_params(**kw), N)
    if p.J == 0: continue
    try:
        pl = tspws.Plan(p, N)
    except tspws.TspwsError as e:
        print("plan", kw, N, e); continue
    X = tspws.synth(mtr, N, seed=seed)
    t0 = time.time(); ls, ts = pl.stack(X); torch.cuda.synchronize(); t1 = time.time()
    want = abi.run_main(main, abi.default_params(**kw), X.cpu().numpy()); t2 = time.time()
    n += 1
    e1, e2 = abi.relerr(ls.cpu().numpy(), want["ls"]), abi.relerr(ts.cpu().numpy(), want["tsPWS"])
    ok = want["rc"] == 0 and e1 < 2e-6 and e2 < 2e-6
    print(("ok      " if ok else "MISMATCH"), seed, kw, "N", N, "mtr", mtr, "J", p.J, "V", p.V, "relerr %.1e %.1e" % (e1, e2), "gpu %.2fs cpu %.1fs" % (t1 - t0, t2 - t1), flush=True)
    bad += 0 if ok else 1
    pl.close()
print("large cases", n, "mismatches", bad)
