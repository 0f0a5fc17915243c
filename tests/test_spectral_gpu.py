"""GPU parity tests of the spectral forward engine (ts-pws_amd/csrc/spectral.hip): the far-decimated octaves of a many-trace batch
through the traces' spectra instead of per-scale FIR sums.  Exact because the reference's decimating FIR is a circular correlation
(FWTa/cdotx.c:35-72, driver FWTa/wavelet_v7.c:43-64): for N a power of two it is the transform's own (those octaves' D divides N), for
any other N it is a linear correlation of the trace's periodic extension, evaluated over a power-of-two window of >= N + L - 1 samples
(outputs k < ceil(N / D), FWTa/wavelet_v7.c:61).  Tolerances as in test_hip_parity.py: 1e-11 on the FP64 coefficients, 2e-6 on the
float32 outputs."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import abi
from conftest import SWEEPS_LIB

pytestmark = pytest.mark.gpu

TOL64 = 1e-11
TOL32 = 2e-6

tspws = importlib.import_module("ts-pws_amd")


@pytest.fixture(scope="module")
def lib():
    lib = tspws.load()
    assert lib.tspws_hip_device_count() > 0, "no MI355X visible: the HIP path cannot run (there is no CPU fallback)"
    return lib


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def dev_forward_spectral(lib, torch, pl, X, nsmax):
    ntr, N = X.shape
    Xd = torch.as_tensor(X, device="cuda")
    Y = torch.zeros((ntr, 2 * pl.ncoef), dtype=torch.float64, device="cuda")
    fn = lib.tspws_hip_forward_spectral_f64 if X.dtype == np.float64 else lib.tspws_hip_forward_spectral_f32
    tspws.check(fn(pl.h, Xd.data_ptr(), ntr, N, Y.data_ptr(), nsmax, None), "forward_spectral")
    torch.cuda.synchronize()
    return Y.cpu().numpy().view(np.complex128)


def test_spectral_coefficients_golden(lib, torch, golden):
    """The reference's own coefficients of one trace (tests/golden/cwt.npz): every scale of the frame's largest spectral set."""
    g = golden["cwt"]
    seen = odd = 0
    for name in sorted({k.split("/")[0] for k in g.files}):
        x = g[f"{name}/x"]
        N = len(x)
        p = abi.default_params()
        p.type, p.J, p.V = int(g[f"{name}/type"]), int(g[f"{name}/J"]), int(g[f"{name}/V"])
        p.s0, p.b0, p.w0 = float(g[f"{name}/s0"]), float(g[f"{name}/b0"]), float(g[f"{name}/w0"])
        pl = tspws.Plan(p, N)
        f = abi.OracleFrame.from_params(p, N)
        sf, se = lib.tspws_hip_spectral_first_scale(pl.h, 1 << 30), lib.tspws_hip_spectral_end_scale(pl.h)
        NT = lib.tspws_hip_spectral_transform_length(pl.h)
        if any(int(d) & (int(d) - 1) for d in f.D):      # (odd decimations: no spectral set)
            assert sf == pl.S, name
            continue
        assert sf < se <= pl.S, name                     # (N = 1501, odd, no decimation divides it: a window of the periodic extension)
        assert NT & (NT - 1) == 0 and (NT == N or NT >= N + int(f.L[se - 1]) - 1), (name, NT)
        off = np.concatenate([[0], np.cumsum(f.Ns.astype(np.int64))])
        want = g[f"{name}/Y"]
        for X in (x[None, :].astype(np.float64), x[None, :].astype(np.float32)):
            Y = dev_forward_spectral(lib, torch, pl, np.ascontiguousarray(X), 1 << 30)[0]
            for s in range(sf, se):
                a, b = int(off[s]), int(off[s + 1])
                assert f.D[s] >= 8 and (N & (N - 1) or N % int(f.D[s]) == 0)
                assert abi.relerr(Y[a:b], want[a:b]) < TOL64, (name, s)
        seen += 1
        odd += N & 1
    assert seen >= 3 and odd >= 1


@pytest.mark.parametrize("kw,N,ntr,nsmax", [
    (dict(), 1024, 3, 1 << 20), (dict(), 4096, 70, 256), (dict(), 131072, 2, 1 << 20), (dict(type=-3), 32768, 5, 4096),
    (dict(w0=2 * np.pi), 32768, 7, 1024), (dict(type=-2), 8192, 2, 1 << 20), (dict(b0=4.0), 65536, 2, 2048), (dict(J=3), 2048, 4, 1 << 20),
    (dict(V=7), 4096, 130, 1 << 20), (dict(s0=4.0, J=5), 8192, 65, 1 << 20), (dict(b0=0.25), 16384, 3, 1 << 20),
    # N not a power of two (the shipped example's 16501 -- five clipped scales stay outside the window --, odd, even, just above / below a power of two)
    (dict(), 16501, 3, 1 << 20), (dict(), 3000, 70, 1 << 20), (dict(type=-3), 20000, 5, 4096), (dict(w0=2 * np.pi), 1501, 66, 1 << 20),
    (dict(), 4097, 3, 1 << 20), (dict(V=5, b0=2.0), 8191, 4, 512), (dict(type=-2), 86400, 2, 2048), (dict(J=6), 1025, 130, 1 << 20),
    # transform windows of 65 536 samples with >= 4 trace blocks: the 64-point first pass (64 x 32 x 16 instead of 16 x 16 x 16 x 8)
    (dict(), 65536, 260, 2048), (dict(type=-3), 40000, 257, 2048),
])
def test_spectral_coefficients_vs_oracle(lib, torch, kw, N, ntr, nsmax):
    """Per-trace coefficients of the spectral set [first, end) for float and double input; one trace carries a stretch of exact zeros."""
    p = abi.resolve(abi.default_params(**kw), N)
    f = abi.OracleFrame.from_params(p, N)
    pl = tspws.Plan(p, N)
    sf, se = lib.tspws_hip_spectral_first_scale(pl.h, nsmax), lib.tspws_hip_spectral_end_scale(pl.h)
    NT = lib.tspws_hip_spectral_transform_length(pl.h)
    assert sf < se <= f.S
    # the set is a run of octaves at the coarse end of what fits the transform window: D >= 8, a power of two, at most nsmax outputs
    assert all(int(f.D[s]) >= 8 and int(f.Ns[s]) <= nsmax for s in range(sf, se))
    assert sf == 0 or int(f.D[sf - 1]) < 8 or int(f.Ns[sf - 1]) > nsmax or se - sf >= 128 - int(p.V)
    if N & (N - 1):
        assert NT & (NT - 1) == 0 and all(N + int(f.L[s]) - 1 <= NT for s in range(se)) and (se == f.S or N + int(f.L[se]) - 1 > NT)
    else:
        assert NT == N and se == f.S
    X = abi.synth_traces(ntr, N, seed=5)
    X[ntr // 2, N // 3: N // 3 + N // 4] = 0
    off = np.concatenate([[0], np.cumsum(f.Ns.astype(np.int64))])
    Y64 = dev_forward_spectral(lib, torch, pl, X.astype(np.float64), nsmax)
    Y32 = dev_forward_spectral(lib, torch, pl, X, nsmax)
    for t in sorted({0, ntr // 2, ntr - 1}):
        Yo = f.forward(X[t].astype(np.float64))
        for s in range(sf, se):
            a, b = int(off[s]), int(off[s + 1])
            assert abi.relerr(Y64[t][a:b], Yo[a:b]) < TOL64, (t, s)
            assert abi.relerr(Y32[t][a:b], Yo[a:b]) < TOL64, (t, s)
        assert not Y64[t][:int(off[sf])].any() and not Y64[t][int(off[se]):].any()   # the other scales are left alone


def test_frames_without_a_spectral_set(lib):
    """N < 1024, decimations 3 * 2^j or 5 * 2^j, a frame that ends before D = 8: TSPWS_E_ARG from the per-trace entry, first scale == S."""
    import torch
    for kw, N in [(dict(), 512), (dict(), 1000), (dict(b0=3.0), 49152), (dict(b0=5.0, J=7), 20480), (dict(J=2), 5000)]:
        p = abi.resolve(abi.default_params(**kw), N)
        pl = tspws.Plan(p, N)
        assert lib.tspws_hip_spectral_first_scale(pl.h, 1 << 30) == pl.S
        X = torch.zeros((2, N), dtype=torch.float64, device="cuda")
        Y = torch.zeros((2, 2 * pl.ncoef), dtype=torch.float64, device="cuda")
        assert lib.tspws_hip_forward_spectral_f64(pl.h, X.data_ptr(), 2, N, Y.data_ptr(), 1 << 30, None) != 0


def test_engine_choice_rule(lib):
    """Default rule (no TSPWS_ENGINE): batches of >= 64 traces and >= 1 M samples (or >= 256 traces) send the octaves with D >= 32 (two-voice
    frames: D >= 16) through the spectrum; smaller batches and frames without a spectral set stay on the FIR kernels."""
    if os.environ.get("TSPWS_ENGINE") or os.environ.get("TSPWS_SPEC_NSMAX"):
        pytest.skip("engine pinned by the environment")
    N = 32768
    p = abi.resolve(abi.default_params(), N)
    pl = tspws.Plan(p, N)
    f = abi.OracleFrame.from_params(p, N)
    s = lib.tspws_hip_spectral_choice(pl.h, 1024)
    assert s < pl.S and int(f.D[s]) == 32 and int(f.D[s - 1]) == 16
    assert lib.tspws_hip_spectral_choice(pl.h, 64) == s              # one full trace block of 2 M samples
    assert lib.tspws_hip_spectral_choice(pl.h, 48) == pl.S           # fewer than 64 traces
    ps = tspws.Plan(abi.resolve(abi.default_params(), 4096), 4096)
    assert lib.tspws_hip_spectral_choice(ps.h, 128) == ps.S          # fewer than 1 M samples and fewer than 256 traces
    assert lib.tspws_hip_spectral_choice(ps.h, 256) < ps.S
    pm = tspws.Plan(abi.resolve(abi.default_params(type=-3), N), N)
    fm = abi.OracleFrame.from_params(abi.resolve(abi.default_params(type=-3), N), N)
    sm = lib.tspws_hip_spectral_choice(pm.h, 1024)                   # Mexican hat (two voices per octave): one octave more, D >= 16
    assert sm < pm.S and int(fm.D[sm]) == 16 and int(fm.D[sm - 1]) == 8
    po = tspws.Plan(abi.resolve(abi.default_params(), 16501), 16501)
    fo = abi.OracleFrame.from_params(abi.resolve(abi.default_params(), 16501), 16501)
    so = lib.tspws_hip_spectral_choice(po.h, 499)                    # the shipped example's shape: odd N, the same octave bound
    assert so < po.S and int(fo.D[so]) == 32 and int(fo.D[so - 1]) == 16
    assert lib.tspws_hip_spectral_choice(po.h, 32) == po.S
    assert lib.tspws_hip_spectral_end_scale(po.h) == po.S - 5 and lib.tspws_hip_spectral_transform_length(po.h) == 32768   # (the five clipped scales)


def _child(env):
    e = dict(os.environ)
    e.update(env)
    if "TSPWS_SPEC_NSMAX" in env:
        e["TSPWS_LIB_PATH"] = SWEEPS_LIB   # (the octave bound is a sweep switch: only the -DTSPWS_SWEEPS build reads it)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "spectral_engine.py")], capture_output=True, text=True,
                       timeout=1500, env=e)
    line = [l for l in r.stdout.splitlines() if l.startswith("SPECTRAL_ENGINE")]
    assert line, r.stdout[-3000:] + r.stderr[-3000:]
    return float(line[0].split()[1]), line[0].split()[2]


def test_whole_calls_with_the_engine_pinned():
    """Every whole-call golden of the reference and ten seeded ensembles (zero traces, stretches of zeros, V = 5 / 7, Mexican hat,
    two-stage with 200 groups) through tspws_main with TSPWS_ENGINE=spectral -- all octaves with D >= 8 spectral -- and with the bound
    of the default rule; the same cases on the FIR kernels agree to the last bit of the float outputs or to 2e-6."""
    e_all, d_all = _child({"TSPWS_ENGINE": "spectral", "TSPWS_SPEC_NSMAX": str(1 << 30)})
    e_def, d_def = _child({"TSPWS_ENGINE": "spectral"})
    e_fir, d_fir = _child({"TSPWS_ENGINE": "fir"})
    assert max(e_all, e_def, e_fir) < TOL32


def test_large_batch_default_rule_matches_fir(lib, torch):
    """1024 x 8192 (the default rule picks the spectral engine) against the same call with the FIR kernels forced through the
    per-trace coefficients: stacks of the spectral scales equal sums of the FIR coefficients."""
    N, ntr = 8192, 1024
    p = abi.resolve(abi.default_params(), N)
    pl = tspws.Plan(p, N)
    f = abi.OracleFrame.from_params(p, N)
    X = tspws.synth(ntr, N, seed=3)
    ls, ts = pl.stack(X)
    torch.cuda.synchronize()
    Xh = X.cpu().numpy()
    sub = 160
    a = abi.run_main(lib.tspws_main, abi.default_params(), Xh[:sub])       # below the many-trace size? (160 x 8192 < 7 M samples: FIR)
    b = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(), Xh[:sub])
    assert abi.relerr(a["tsPWS"], b["tsPWS"]) < TOL32 and abi.relerr(a["ls"], b["ls"]) < TOL32
    # full batch: linear stack against the frame-filtered mean through the oracle's transform of the FP64 mean trace (linearity)
    m = Xh.astype(np.float64).sum(axis=0)
    lin = f.inverse(f.forward(m)) / ntr
    assert abi.relerr(ls.cpu().numpy(), lin.astype(np.float32)) < TOL32


@pytest.mark.parametrize("kw,mtr,N,n,d", [(dict(type=-3, Kmax=10), 400, 4096, 10, 1), (dict(Kmax=8, unbiased=1), 300, 8192, 9, 1), (dict(Kmax=12, wu=1.0), 500, 2048, 6, 2)])
def test_jackknife_rows_through_the_spectral_engine(lib, kw, mtr, N, n, d):
    """The one-pass stack + jackknife call transforms (replicas + 1) x Kmax partial stacks: from 64 rows on, the octaves with D >= 16 of
    those rows go through the spectral engine (lanes = rows, per-column stacks by k_spec_stack_rows) -- 110 rows (n = 10, d = 1),
    80 rows, 192 rows -- against the oracle's tspws_main, replicas included; one bin of traces is all zero."""
    Cn = abi.binomial(n, d)
    assert (Cn + 1) * kw["Kmax"] >= 64
    X = abi.synth_traces(mtr, N, seed=71)
    rng = np.random.default_rng(11)
    times = (1262304000 + 86400 * np.sort(rng.integers(0, 365, mtr))).astype(np.int64)
    X[5] = 0
    X[7, N // 2:] = 0
    p = abi.default_params(jackknife_n=n, jackknife_d=d, **kw)
    a = abi.run_main(lib.tspws_main, p, X, times=times)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X, times=times)
    assert a["rc"] == 0 and b["rc"] == 0
    assert abi.relerr(a["ls"], b["ls"]) < TOL32 and abi.relerr(a["tsPWS"], b["tsPWS"]) < TOL32
    np.testing.assert_array_equal(a["jk_mtr"], b["jk_mtr"])
    for c in range(Cn):
        assert abi.relerr(a["jk_ts"][c], b["jk_ts"][c]) < TOL32 and abi.relerr(a["jk_ls"][c], b["jk_ls"][c]) < TOL32, c


@pytest.mark.parametrize("N", [4096, 4097])
def test_spectral_engine_over_several_batches(sweeps, torch, monkeypatch, N):
    """Large ensembles are walked in batches (transposed copy <= 1 GiB, at most 4096 traces): forced here to 64 / 128 traces per batch
    (TSPWS_TL_BATCH, sweeps build) on 300 traces -- the later batches add their blocks' planes to the stacks of the first, the last batch is
    partial; the engine is the default rule's (300 traces: spectral) and the pinned one.  N = 4097: the window of the periodic extension and
    the dense contraction of the clipped scales (fwd_gemm.h), whose run buffer is reused batch after batch."""
    sw, swlib = sweeps
    mtr = 300
    X = abi.synth_traces(mtr, N, seed=81)
    X[17] = 0
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(wu=1.5), X)
    pl = sw.Plan(sw.resolve(abi.default_params(wu=1.5), N), N)
    assert swlib.tspws_hip_spectral_choice(pl.h, mtr) < pl.S
    for b in ("64", "128", None):
        if b is None:
            monkeypatch.delenv("TSPWS_TL_BATCH", raising=False)
        else:
            monkeypatch.setenv("TSPWS_TL_BATCH", b)
        ls, ts = pl.stack(torch.as_tensor(X, device="cuda"))
        torch.cuda.synchronize()
        assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32, b


def test_one_plan_serves_small_and_large_batches(lib, torch):
    """The fold's class count is sized for the batch: a plan that first sees 3 trace blocks and then 20 (and then 3 again) builds two
    decompositions of the same spectral set and keeps both."""
    N = 8192
    p = abi.default_params()
    pl = tspws.Plan(tspws.resolve(p, N), N)
    for mtr, seed in ((160, 1), (1270, 2), (130, 3)):
        X = abi.synth_traces(mtr, N, seed=90 + seed)
        ls, ts = pl.stack(torch.as_tensor(X, device="cuda"))
        torch.cuda.synchronize()
        want = abi.run_main(abi.oracle().orc_tspws_main, p, X)
        assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32, mtr
