#!/usr/bin/env python3
"""Two calls in flight: the finish stage of call i (stream B) beside the streaming stage of call i+1 (stream A).
Two plans alternate so that every call in flight has its own P / coefficient blocks."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd")
N, mtr, K = 131072, 10000, 10
p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
plans = [tspws.Plan(p, N), tspws.Plan(p, N)]
X = tspws.synth(mtr, N, seed=1)
ls = [torch.empty(N, dtype=torch.float32, device="cuda") for _ in range(2)]
ts = [torch.empty(N, dtype=torch.float32, device="cuda") for _ in range(2)]
A = torch.cuda.Stream(); B = torch.cuda.Stream(priority=int(os.environ.get("B_PRIO", "0")))
done = [torch.cuda.Event(), torch.cuda.Event()]
ready = [torch.cuda.Event(), torch.cuda.Event()]

def serial(n):
    with torch.cuda.stream(A):
        for i in range(n):
            plans[0].stack_local(X, 0, mtr); plans[0].stack_finish(mtr, ls[0], ts[0])

def piped(n):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(A):
            if i >= 2: A.wait_event(done[k])          # P of this plan is free again
            plans[k].stack_local(X, 0, mtr)
            ready[k].record(A)
        with torch.cuda.stream(B):
            B.wait_event(ready[k])
            plans[k].stack_finish(mtr, ls[k], ts[k])
            done[k].record(B)

def timeit(fn, n=30):
    fn(4); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("serial   %.4f ms/step" % timeit(serial))
    print("piped    %.4f ms/step" % timeit(piped))
ref_ls, ref_ts = ls[0].clone(), ts[0].clone()
serial(1); torch.cuda.synchronize()
print("same outputs:", torch.equal(ref_ls, ls[0]), torch.equal(ref_ts, ts[0]), torch.equal(ls[0], ls[1]))
