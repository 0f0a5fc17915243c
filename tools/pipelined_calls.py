#!/usr/bin/env python3
"""Back-to-back calls on ONE GPU with two plans on two streams: the finish stage of call i (FP64-bound, 0.24 ms) runs beside the
streaming stage of call i + 1 (HBM-bound, 0.73 ms) -- what a host that stacks many ensembles of one shape can do with the two
halves of the call (tspws_hip_stack_local / tspws_hip_stack_finish).  Prints the per-call time of the plain loop and of the
pipelined loop and checks that both produce the same outputs.   usage: pipelined_calls.py [mtr] [N] [calls]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd"); tspws.load()
mtr = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
p = tspws.resolve(abi.default_params(Kmax=10, unbiased=1), N)
X = tspws.synth(mtr, N, seed=1)
plans = [tspws.Plan(p, N), tspws.Plan(p, N)]
ls = [torch.empty(N, dtype=torch.float32, device="cuda") for _ in range(2)]
ts = [torch.empty(N, dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def plain(n):
    for i in range(n):
        plans[0].stack_single(X, ls[0], ts[0])


def piped(n):
    prev = None
    for i in range(n):
        s, pl = streams[i % 2], plans[i % 2]
        with torch.cuda.stream(s):
            if prev is not None:
                s.wait_event(prev)          # one streaming stage at a time: they would only share the HBM
            pl.stack_local(X, 0, mtr)
            prev = torch.cuda.Event()
            prev.record(s)
            pl.stack_finish(mtr, ls[i % 2], ts[i % 2])


for fn, name in ((plain, "plain loop"), (piped, "two plans, two streams"), (plain, "plain loop"), (piped, "two plans, two streams")):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{name:24s}: {dt * 1e3:.3f} ms/call, {mtr * N / dt:.3e} samples/s")
print("outputs equal:", bool(torch.equal(ls[0], ls[1]) and torch.equal(ts[0], ts[1])))
