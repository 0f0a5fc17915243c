"""Child process of tests/test_spectral_gpu.py: whole tspws_main calls with the forward engine pinned by TSPWS_ENGINE (read once per
process by the library) -- every single-stage / two-stage golden of the reference (16 x 2048: with TSPWS_ENGINE=spectral even these
small ensembles take the many-trace path and the spectral engine) and seeded ensembles against the oracle, among them all-zero
traces and traces with stretches of exact zeros (the reference skips 0 / 0 in the phase stack, ts_pws1f_lib.c:491-492).
Prints SPECTRAL_ENGINE <worst relative error> <digest of all float outputs>."""
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abi
from test_oracle_vs_golden import check_main, main_case_names

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
worst = 0.0
h = hashlib.sha1()

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mains.npz"), allow_pickle=False)
for name in main_case_names(g):
    r = check_main(lib.tspws_main, g, name, 2e-6)
    worst = max(worst, abi.relerr(r["ls"], g[f"{name}/ls"]), abi.relerr(r["tsPWS"], g[f"{name}/tsPWS"]))
    h.update(r["ls"].tobytes()); h.update(r["tsPWS"].tobytes())


def check(kw, mtr, N, seed, holes=True):
    global worst
    X = abi.synth_traces(mtr, N, seed=seed)
    if holes:
        X[mtr // 3] = 0                                   # an all-zero trace
        X[mtr // 2, N // 4: N // 2] = 0                   # a stretch of exact zeros longer than most filters
        X[mtr - 1, : N // 8] = 0
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["rc"] == 0 and b["rc"] == 0, (a["rc"], b["rc"])
    e = max(abi.relerr(a["ls"], b["ls"]), abi.relerr(a["tsPWS"], b["tsPWS"]))
    assert e < 2e-6, (kw, mtr, N, e)
    worst = max(worst, e)
    h.update(a["ls"].tobytes()); h.update(a["tsPWS"].tobytes())


check(dict(), 100, 4096, 1)
check(dict(w0=2 * np.pi), 200, 8192, 2)                  # V = 5
check(dict(type=-3), 70, 4096, 3)                        # Mexican hat: two voices per octave
check(dict(type=-2, unbiased=1), 130, 2048, 4)           # exact Morlet, unbiased
check(dict(wu=1.5), 64, 1024, 5)                         # the smallest frame with a spectral set
check(dict(V=7, J=6), 90, 4096, 6)
check(dict(b0=4.0), 66, 16384, 7)                        # D = 8 .. : every octave spectral when the bound allows
check(dict(lrm=1, wu=1.0), 257, 2048, 8)                 # five trace blocks, the last one with a single trace
check(dict(), 40, 3000, 9, holes=False)                  # N not a power of two: a window of the periodic extension (partially filled trace block)
check(dict(), 100, 3000, 11)
check(dict(type=-3), 70, 5000, 12)                       # Mexican hat, N even but no power of two
check(dict(w0=2 * np.pi, wu=1.0), 130, 1501, 13)         # odd N: no decimation divides it (seam outputs k = ceil(N / D) - 1)
check(dict(), 66, 16501, 14)                             # the shipped example's length: five clipped scales stay on the direct kernel
check(dict(Kmax=70, unbiased=1), 300, 4097, 15)          # two-stage with 70 groups just above a power of two: the window is nearly twice the trace
# the shipped example's first 32 traces (N = 16501) against the reference's own outputs
ge = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example32.npz"), allow_pickle=False)
from test_oracle_vs_golden import _case_params
for name in ("ex1", "ex2", "ex3", "ex_mexhat"):
    r = abi.run_main(lib.tspws_main, _case_params(ge, name), ge["traces"], dt=float(ge["dt"]), beg=float(ge["beg"]))
    e = max(abi.relerr(r["ls"], ge[f"{name}/ls"]), abi.relerr(r["tsPWS"], ge[f"{name}/tsPWS"]))
    assert r["rc"] == 0 and e < 2e-6, (name, e)
    worst = max(worst, e)
    h.update(r["ls"].tobytes()); h.update(r["tsPWS"].tobytes())
check(dict(Kmax=200, unbiased=1), 600, 2048, 10)         # two-stage with 200 groups: the partial stacks are a many-trace batch (double input)


# Ensembles that live AT the engine's noise floor (a coefficient at or below the transforms' rounding noise is skipped by the
# phase stack like the reference's exact zero, ts_pws1f_lib.c:489-492): constant traces, band-limited traces (part of the frame sees rounding
# noise only), a spike over a 1e-13 background -- tools/noise_floor_probe.py measured 0.0 on the float outputs with either engine
def check_traces(kw, X, what):
    global worst
    X = np.ascontiguousarray(X.astype(np.float32))
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    e = max(abi.relerr(a["ls"], b["ls"]), abi.relerr(a["tsPWS"], b["tsPWS"]))
    assert a["rc"] == 0 and b["rc"] == 0 and e < 2e-6, (what, kw, e)
    worst = max(worst, e)
    h.update(a["ls"].tobytes()); h.update(a["tsPWS"].tobytes())


rng = np.random.default_rng(7)
mtr, N = 96, 4096
noise = rng.uniform(-0.5, 0.5, (mtr, N))
F = np.fft.rfft(noise, axis=1)
lo, hi = F.copy(), F.copy()
lo[:, N // 16:] = 0                                      # low-pass: the fine scales see rounding noise only
hi[:, :N // 8] = 0                                       # high-pass: the far-decimated (spectral) scales see rounding noise only
spike = 1e-13 * noise
spike[:, N // 2] += 1.0
for kw in (dict(), dict(type=-2), dict(Kmax=70, unbiased=1)):
    check_traces(kw, np.repeat(rng.uniform(0.5, 2.0, (mtr, 1)), N, axis=1), "DC only")
    check_traces(kw, 50.0 + noise, "DC + noise")
    check_traces(kw, np.fft.irfft(lo, N, axis=1), "low-pass")
    check_traces(kw, np.fft.irfft(hi, N, axis=1), "high-pass")
    check_traces(kw, spike, "spike over a 1e-13 background")
print("SPECTRAL_ENGINE", worst, h.hexdigest()[:16])
