#!/usr/bin/env python3
"""What the spectral engine's noise floor (a coefficient at or below the transforms' rounding noise is skipped by the phase stack like
the reference's exact zero) does to ensembles that LIVE at that floor: DC-only traces, band-limited traces (no energy in part of the frame), a spike
over a 1e-13 background.  Whole tspws_main calls, engine pinned by TSPWS_ENGINE (run once with fir, once with spectral), against the oracle."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()


def cases(mtr=96, N=4096):
    rng = np.random.default_rng(7)
    noise = rng.uniform(-0.5, 0.5, (mtr, N))
    F = np.fft.rfft(noise, axis=1)
    lo, hi = F.copy(), F.copy()
    lo[:, N // 16:] = 0          # low-pass: nothing above 1/16 of Nyquist -> the fine scales see rounding noise only
    hi[:, :N // 8] = 0           # high-pass: the far-decimated (spectral) scales see rounding noise only
    spike = 1e-13 * noise
    spike[:, N // 2] += 1.0
    return {
        "dc_only": np.repeat(rng.uniform(0.5, 2.0, (mtr, 1)), N, axis=1),
        "dc_plus_noise": 50.0 + noise,
        "low_pass": np.fft.irfft(lo, N, axis=1),
        "high_pass": np.fft.irfft(hi, N, axis=1),
        "spike_over_1e-13": spike,
    }


if __name__ == "__main__":
    for kw in (dict(), dict(type=-2), dict(Kmax=70, unbiased=1)):
        for name, X in cases().items():
            X = np.ascontiguousarray(X.astype(np.float32))
            p = abi.default_params(**kw)
            a = abi.run_main(lib.tspws_main, p, X)
            b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
            print(f"{os.environ.get('TSPWS_ENGINE', 'auto'):9s} {str(kw):32s} {name:18s} rc {a['rc']} {b['rc']}  relerr ls {abi.relerr(a['ls'], b['ls']):.2e} tsPWS {abi.relerr(a['tsPWS'], b['tsPWS']):.2e}"
                  f"   max|ls| {np.abs(b['ls']).max():.2e} max|tsPWS| {np.abs(b['tsPWS']).max():.2e}")
