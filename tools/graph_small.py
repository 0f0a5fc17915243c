#!/usr/bin/env python3
"""Small calls on a stream of the caller's own: launch by launch (TSPWS_GRAPH=0) against the captured graph (default), same process
cannot switch (the flag is read once), so run it twice.  Checks the replayed graph's outputs against a plan that never captured.
usage: graph_small.py [mtr] [N]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd"); tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
X = tspws.synth(mtr, N, seed=1)
X2 = tspws.synth(mtr, N, seed=2)
st = torch.cuda.Stream()
for name, kw in (("two-stage K=10 unbiased", dict(Kmax=10, unbiased=1)), ("single-stage", dict())):
    p = tspws.resolve(abi.default_params(**kw), N)
    pl, ref = tspws.Plan(p, N), tspws.Plan(p, N)
    ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
    Xw = X.clone()
    with torch.cuda.stream(st):
        for _ in range(4):
            pl.stack_single(Xw, ls, ts)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pl.stack_single(Xw, ls, ts)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 200
        # the graph reads the traces at call time: new contents in the same buffer
        Xw.copy_(X2)
        pl.stack_single(Xw, ls, ts)
        st.synchronize()
    want = ref.stack_single(X2)          # default stream: never captured
    torch.cuda.synchronize()
    ok = bool(torch.equal(want[0], ls) and torch.equal(want[1], ts))
    print(f"{mtr} x {N} {name}: {dt * 1e3:.4f} ms/call on a side stream (TSPWS_GRAPH={os.environ.get('TSPWS_GRAPH', '1')}), outputs equal to the uncaptured plan: {ok}")
