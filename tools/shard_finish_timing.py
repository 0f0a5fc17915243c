#!/usr/bin/env python3
"""Finish stage per rank when the scales are sharded over `world` ranks (tspws_hip_stack_finish_scales): compute time of
every rank's share on ONE GPU (the 2 N-double all-reduce between the ranks is not included), next to the plain finish."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, abi
tspws = importlib.import_module("ts-pws_amd")
N, mtr, K = 131072, 10000, 10
pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N), N)
X = tspws.synth(mtr, N, seed=1)
ls = torch.empty(N, dtype=torch.float32, device="cuda"); ts = torch.empty(N, dtype=torch.float32, device="cuda")
x2 = torch.empty(2 * N, dtype=torch.float64, device="cuda")
pl.stack_local(X, 0, mtr)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("plain finish            %.4f ms" % timeit(lambda: pl.stack_finish(mtr, ls, ts)))
for world in (2, 4, 8):
    shares = [pl.finish_shard(mtr, r, world) for r in range(world)]
    t = [timeit(lambda a=a, b=b: (pl.stack_finish_scales(mtr, a, b, x2), pl.epilogue(x2, mtr, ls, ts))) for a, b in shares]
    print("world %d shares %s" % (world, shares))
    print("        per rank ms: %s   max %.4f" % (" ".join("%.3f" % v for v in t), max(t)))
# per-octave times (one share = one decimation octave), for tuning the cost model of tspws_hip_finish_shard
D = pl.tables()["D"]
octs = [0] + [s for s in range(1, pl.S) if D[s] != D[s - 1]] + [pl.S]
print("per octave ms:", " ".join("D=%d:%.3f" % (D[a], timeit(lambda a=a, b=b: pl.stack_finish_scales(mtr, a, b, x2), 20)) for a, b in zip(octs[:-1], octs[1:])))
