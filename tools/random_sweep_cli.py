#!/usr/bin/env python3
"""Seeded sweep over the SAC command line (ts-pws_amd/bin/ts_pws): random ensembles written as SAC files (a symmetric lag
axis or not, a few traces of another sampling interval that the reader must skip), random options, with and without the
intermediate `bin` container -- against tspws_main called directly on the same traces.
usage: random_sweep_cli.py [first_seed [n_seeds]]"""
import importlib, os, struct, subprocess, sys, tempfile
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, abi
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
EXE = os.path.join(R, "ts-pws_amd", "bin", "ts_pws")
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 25
bad = n = nbin = nfold = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(3000 + seed)
    for it in range(3):
        N = int(rng.choice([501, 800, 1001, 1500, 2048])); mtr = int(rng.choice([3, 9, 20, 41]))
        dt = float(rng.choice([1.0, 0.5, 4.0]))
        sym = rng.random() < 0.5
        beg = -0.5 * (N - 1) * dt if sym else float(rng.choice([0.0, -10.0, 3.0]))
        X = abi.synth_traces(mtr, N, seed=seed * 5 + it)
        args, kw = [], {}
        if rng.random() < 0.3: args.append("MexHat"); kw["type"] = -3
        elif rng.random() < 0.4:
            w0 = float(np.round(rng.uniform(4.0, 11.0), 3)); args.append(f"w0={w0}"); kw["w0"] = w0
        if rng.random() < 0.4:
            K = int(rng.integers(1, 8)); args.append(f"TwoStage={K}"); kw["Kmax"] = K
        wu = float(rng.choice([2.0, 1.0, 1.5])); args.append(f"wu={wu}"); kw["wu"] = wu
        if wu == 2.0 and rng.random() < 0.5: args.append("unbiased"); kw["unbiased"] = 1
        if rng.random() < 0.4: args.append("rm"); kw["lrm"] = 1
        fold = rng.random() < 0.4
        if fold: args.append("fold"); kw["fold"] = 1
        if rng.random() < 0.3:
            J = int(rng.integers(2, 6)); args.append(f"J={J}"); kw["J"] = J
        if rng.random() < 0.25:
            nm = int(rng.integers(1, mtr + 1)); args.append(f"Nmax={nm}"); kw["Nmax"] = nm
        use_bin = rng.random() < 0.4
        with tempfile.TemporaryDirectory() as d:
            names = []
            for i in range(mtr):
                pth = os.path.join(d, f"t{i:03d}.sac")
                abi.write_sac(pth, X[i], dt, beg, year=2011, jday=1 + (7 * i) % 360, kstnm="ST%d" % (i % 3))
                names.append(pth)
            open(os.path.join(d, "list.txt"), "w").write("\n".join(names) + "\n")
            src = ["list.txt"]
            if use_bin:
                # msacs container (sac2bin.h:6-27 / sac2bin.c:192-195): 116-byte header, time_t[mtr], float lag0[mtr], float data[mtr][n]
                hdr = struct.pack("<8s8s8s8s8s8s8s8s8s6fII3f", b"ccgn", b"", b"ECH", b"00", b"Z", b"", b"CAN", b"00", b"Z",
                                  48.2, 7.15, 0.0, -35.3, 149.0, 0.0, N, mtr, dt * (N - 1), beg, beg + dt * (N - 1))
                tms = (1293840000 + 86400 * ((7 * np.arange(mtr)) % 360)).astype(np.int64)
                with open(os.path.join(d, "ens.bin"), "wb") as f:
                    f.write(hdr + tms.tobytes() + np.zeros(mtr, np.float32).tobytes() + X.astype(np.float32).tobytes())
                src = ["ens.bin", "bin"]
            r = subprocess.run([EXE, *src, "osac=o", *args], cwd=d, capture_output=True, text=True, timeout=300)
            n += 1
            nbin += int(use_bin)
            p = abi.default_params(**kw)
            a = abi.run_main(lib.tspws_main, p, X, dt=dt, beg=beg)
            nfold += int(bool(a["params"].fold))
            msgs = []
            if r.returncode != 0: msgs.append(f"cli rc {r.returncode}: {r.stdout[-200:]} {r.stderr[-200:]}")
            else:
                try:
                    ts = abi.read_sac(os.path.join(d, "ts_pws_o.sac")); ls = abi.read_sac(os.path.join(d, "tl_o.sac"))
                    folded = bool(a["params"].fold)
                    h = N // 2 if folded else 0        # folded outputs keep samples max/2 .. (ts_pws1f.c:322-328)
                    if len(ts["data"]) != N - h: msgs.append(f"npts {len(ts['data'])} vs {N - h}")
                    elif not (np.array_equal(ts["data"], a["tsPWS"][h:]) and np.array_equal(ls["data"], a["ls"][h:])): msgs.append("data differ")
                    exp_b = beg + dt * h if folded else beg
                    if abs(float(ts["f"][5]) - np.float32(exp_b)) > 1e-3 * max(1.0, abs(exp_b)): msgs.append(f"b {ts['f'][5]} vs {exp_b}")
                    exp_m = kw.get("Nmax", mtr)
                    if int(round(float(ts["f"][40]))) != exp_m: msgs.append(f"user0 {ts['f'][40]} vs {exp_m}")
                except FileNotFoundError as e:
                    msgs.append(f"missing output {e}")
            if msgs:
                bad += 1
                print("MISMATCH", seed, it, args, "bin" if use_bin else "sac", "N", N, "mtr", mtr, "dt", dt, "beg", beg, msgs, flush=True)
print("cli cases", n, "through the bin container", nbin, "folded", nfold, "mismatches", bad)
