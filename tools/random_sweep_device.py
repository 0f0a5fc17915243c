#!/usr/bin/env python3
"""Seeded sweep over the device-resident / sharded entry points: random trace shards (empty ones included), padded rows,
finish in pieces, scale-sharded finish with random world sizes -- against the one-shot call and the oracle.
usage: random_sweep_device.py [first_seed [n_seeds]]"""
import importlib, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, abi
import test_hip_parity as T
tspws = importlib.import_module("ts-pws_amd"); lib = tspws.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = n = 0
def f32(t): return t.cpu().numpy()
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(9000 + seed)
    for it in range(4):
        kw, N, mtr, beg = T._random_case(rng)
        for k in ("fold", "lrm"): kw.pop(k, None)
        mtr = int(rng.choice([5, 20, 64, 100, 130, 300])); N = int(rng.choice([509, 1024, 2048, 3001, 4096, 8192]))
        if rng.random() < 0.7: kw["Kmax"] = int(rng.integers(1, 13))
        p = tspws.resolve(abi.default_params(**kw), N)
        if p.J == 0: continue
        pad = int(rng.choice([0, 0, 1, 4, 12]))
        try:
            pl = tspws.Plan(p, N)
        except tspws.TspwsError:
            continue
        X = tspws.synth(mtr, N, seed=seed * 10 + it, pad=pad)
        ls0, ts0 = pl.stack(X)
        torch.cuda.synchronize()
        msgs = []
        want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), f32(X).copy())
        if want["rc"] != 0: continue
        if abi.relerr(f32(ls0), want["ls"]) >= 2e-6 or abi.relerr(f32(ts0), want["tsPWS"]) >= 2e-6: msgs.append("one-shot vs oracle")
        # trace shards
        cuts = sorted(set([0, mtr] + [int(c) for c in rng.integers(0, mtr + 1, int(rng.integers(1, 4)))]))
        if rng.random() < 0.3: cuts = sorted(cuts + [cuts[1]])       # an empty shard
        tot = torch.zeros_like(pl.reduce_buffer(mtr))
        for a, b in zip(cuts[:-1], cuts[1:]):
            pl.stack_local(X[a:b], a, mtr); torch.cuda.synchronize(); tot += pl.reduce_buffer(mtr)
        pl.reduce_buffer(mtr).copy_(tot)
        ls1 = torch.empty_like(ls0); ts1 = torch.empty_like(ts0)
        two = bool(p.Kmax) and p.Kmax <= mtr
        if two and p.Kmax >= 2 and rng.random() < 0.5:               # finish in pieces
            h = int(rng.integers(1, p.Kmax))
            pl.stack_finish_range(mtr, 0, h); pl.stack_finish_range(mtr, h, p.Kmax); pl.stack_finish_tail(mtr, ls1, ts1)
        else:
            pl.stack_finish(mtr, ls1, ts1)
        torch.cuda.synchronize()
        if abi.relerr(f32(ls1), f32(ls0)) >= 1e-6 or abi.relerr(f32(ts1), f32(ts0)) >= 1e-6: msgs.append(f"trace shards {cuts}")
        # scale shards
        if two:
            world = int(rng.integers(2, 7))
            shares = [pl.finish_shard(mtr, r, world) for r in range(world)]
            if all(sh is not None for sh in shares):
                x2 = torch.empty(2 * N, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(x2)
                for a, b in shares:
                    pl.stack_finish_scales(mtr, a, b, x2); acc += x2
                ls2 = torch.empty_like(ls0); ts2 = torch.empty_like(ts0)
                pl.epilogue(acc, mtr, ls2, ts2); torch.cuda.synchronize()
                cov = [s for a, b in shares for s in range(a, b)]
                if cov != list(range(pl.S)): msgs.append(f"shares do not partition: {shares}")
                if abi.relerr(f32(ls2), f32(ls0)) >= 1e-6 or abi.relerr(f32(ts2), f32(ts0)) >= 1e-6: msgs.append(f"scale shards {shares}")
        n += 1
        if msgs:
            bad += 1
            print("MISMATCH", seed, it, kw, "N", N, "mtr", mtr, "pad", pad, msgs, flush=True)
        pl.close()
print("device cases", n, "mismatches", bad)
