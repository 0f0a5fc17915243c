"""GPU parity tests: the HIP path (through the C-ABI of include/tspws_hip.h and the drop-in
tspws_main) against the oracle and the golden vectors generated from the reference.

Tolerance: north_star asks for 1e-5 relative (max|a-b| / max|b|) on the float32 outputs; the
device path is FP64 like the reference, so the tests hold it to much tighter bounds:
  * FP64 intermediates (taps, coefficients, reconstructions): 1e-11
  * float32 outputs: 2e-6 (an ulp of the peak, plus rounding of values near a float tie)
"""
import ctypes as C
import os
import importlib

import numpy as np
import pytest

import abi
from test_oracle_vs_golden import check_main, main_case_names, _case_params
from conftest import SWEEPS_LIB

pytestmark = pytest.mark.gpu

TOL64 = 1e-11
TOL32 = 2e-6
NORTH_STAR_TOL = 1e-5

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


def plan_for(params, N):
    return tspws.Plan(tspws.resolve(params, N), N)


# ---------------------------------------------------------------------------- frame
@pytest.mark.parametrize("kw,N", [
    (dict(), 2048), (dict(), 16501), (dict(w0=2 * np.pi), 32768), (dict(type=-3), 2048),
    (dict(type=-2), 2048), (dict(s0=3.7, J=4), 1501), (dict(uni=1, J=3), 2048), (dict(), 131072), (dict(type=-3), 131072),
])
def test_frame_matches_oracle(lib, kw, N):
    p = abi.resolve(abi.default_params(**kw), N)
    f = abi.OracleFrame.from_params(p, N)
    pl = tspws.Plan(p, N)
    t = pl.tables()
    assert pl.S == f.S and pl.ncoef == f.ncoef and pl.ntaps == f.ntaps
    for k in ("L", "c", "cd", "D", "Ns"):
        np.testing.assert_array_equal(t[k], getattr(f, k))
    np.testing.assert_array_equal(t["scale"], f.scale)
    assert pl.Cpsi == f.Cpsi
    w, wd = pl.taps()
    ow, owd = f.taps()
    # device libm vs glibc: a few ulp on sincos/exp
    assert np.max(np.abs(w - ow)) <= 1e-14 * np.max(np.abs(ow))
    assert np.max(np.abs(wd - owd)) <= 1e-14 * np.max(np.abs(owd))


def test_frame_known_answers(lib):
    pl = plan_for(abi.default_params(), 131072)
    assert pl.info.J == 14 and pl.S == 56 and pl.ncoef == 524256 and pl.ntaps == 1383505
    assert pl.Cpsi == pytest.approx(0.29982027317664373, rel=1e-15)
    w, _ = pl.taps()
    assert w[10] == pytest.approx(0.53112596601359841 + 0j, rel=1e-13)


# ---------------------------------------------------------------------- transforms
def dev_forward(torch, pl, x):
    xd = torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    if xd.ndim == 1:
        xd = xd[None, :]
    Y = torch.zeros((xd.shape[0], pl.ncoef, 2), dtype=torch.float64, device="cuda")
    fn = pl.lib.tspws_hip_forward_f64 if xd.dtype == torch.float64 else pl.lib.tspws_hip_forward_f32
    tspws.check(fn(pl.h, xd.data_ptr(), xd.shape[0], xd.shape[1], Y.data_ptr(), None), "forward")
    torch.cuda.synchronize()
    return Y.cpu().numpy().view(np.complex128)[..., 0]


def dev_inverse(torch, pl, Y):
    Y = np.ascontiguousarray(np.atleast_2d(Y), np.complex128)
    Yd = torch.as_tensor(Y.view(np.float64), device="cuda")
    x = torch.zeros((Y.shape[0], pl.N), dtype=torch.float64, device="cuda")
    tspws.check(pl.lib.tspws_hip_inverse(pl.h, Yd.data_ptr(), Y.shape[0], x.data_ptr(), None), "inverse")
    torch.cuda.synchronize()
    return x.cpu().numpy()


def test_forward_inverse_golden(lib, torch, golden):
    g = golden["cwt"]
    for name in sorted({k.split("/")[0] for k in g.files}):
        x = g[f"{name}/x"]
        p = abi.default_params()
        p.type, p.J, p.V = int(g[f"{name}/type"]), int(g[f"{name}/J"]), int(g[f"{name}/V"])
        p.s0, p.b0, p.w0 = float(g[f"{name}/s0"]), float(g[f"{name}/b0"]), float(g[f"{name}/w0"])
        pl = tspws.Plan(p, len(x))
        Y = dev_forward(torch, pl, x)[0]
        assert abi.relerr(Y, g[f"{name}/Y"]) < TOL64, name
        # float32 input path gives the same coefficients for float-representable samples
        Yf = dev_forward(torch, pl, x.astype(np.float32))[0]
        assert abi.relerr(Yf, g[f"{name}/Y"]) < TOL64, name
        xr = dev_inverse(torch, pl, g[f"{name}/Y"])[0]
        assert abi.relerr(xr, g[f"{name}/xrec"]) < TOL64, name


@pytest.mark.parametrize("kw,N,ntr", [
    (dict(), 4096, 3), (dict(), 16501, 2), (dict(type=-3), 8192, 2), (dict(w0=2 * np.pi), 32768, 1),
    (dict(s0=3.7, J=5), 3001, 2), (dict(uni=1, J=4), 1024, 2), (dict(), 131072, 1), (dict(type=-3), 131072, 1),
    (dict(J=2), 64, 3), (dict(s0=7.890778, J=3), 16501, 1),
    (dict(b0=0.25), 32768, 2),   # 40-67 taps per phase: the LDS kernel's tiled (non-resident) path at D >= 64
    (dict(b0=0.25, type=-3, V=3), 8192, 3), (dict(b0=4.0), 65536, 2),
    # decimations 3 * 2^j / 5 * 2^j that divide N: 64-phase chunks with idle lanes (D = 96, 80, 160 ...) on the polyphase kernels
    (dict(b0=3.0), 49152, 2), (dict(b0=5.0, J=7), 81920, 1), (dict(b0=1.5, type=-3), 12288, 3),
])
def test_forward_inverse_vs_oracle(lib, torch, kw, N, ntr):
    p = abi.resolve(abi.default_params(**kw), N)
    f = abi.OracleFrame.from_params(p, N)
    pl = tspws.Plan(p, N)
    X = abi.synth_traces(ntr, N, seed=21).astype(np.float64)
    Y = dev_forward(torch, pl, X)
    Yo = np.stack([f.forward(x) for x in X])
    assert abi.relerr(Y, Yo) < TOL64
    # per-scale check so that a small scale cannot hide behind a large one
    off = np.concatenate([[0], np.cumsum(f.Ns)]).astype(np.int64)
    for s in range(f.S):
        assert abi.relerr(Y[:, off[s]:off[s + 1]], Yo[:, off[s]:off[s + 1]]) < 1e-10, s
    xr = dev_inverse(torch, pl, Yo)
    xo = np.stack([f.inverse(y) for y in Yo])
    assert abi.relerr(xr, xo) < TOL64
    # linearity of the transform pair (size-independent property)
    if ntr >= 2:
        Ysum = dev_forward(torch, pl, (X[0] + 2 * X[1])[None, :])[0]
        assert abi.relerr(Ysum, Y[0] + 2 * Y[1]) < 1e-12


def test_accumulate_and_weight(lib, torch):
    N = 2048
    p = abi.resolve(abi.default_params(), N)
    f = abi.OracleFrame.from_params(p, N)
    pl = tspws.Plan(p, N)
    rng = np.random.default_rng(2)
    Y = (rng.standard_normal((5, f.ncoef)) + 1j * rng.standard_normal((5, f.ncoef)))
    Y[2, ::7] = 0.0  # exact zeros: the NaN-skip rule of the phase stack
    Y[3, 5] = 1e-310 + 0j  # subnormal
    ST = np.zeros(f.ncoef, np.complex128)
    PS = np.zeros(f.ncoef, np.complex128)
    orc = abi.oracle()
    for y in Y:
        yy = np.ascontiguousarray(y)
        orc.orc_accumulate(ST.ctypes.data, PS.ctypes.data, yy.ctypes.data, f.ncoef)
    Yd = torch.as_tensor(Y.view(np.float64), device="cuda")
    STd = torch.zeros(2 * f.ncoef, dtype=torch.float64, device="cuda")
    PSd = torch.zeros_like(STd)
    tspws.check(lib.tspws_hip_accumulate(pl.h, Yd.data_ptr(), 3, STd.data_ptr(), PSd.data_ptr(), 1, None))
    tspws.check(lib.tspws_hip_accumulate(pl.h, Yd[3:].data_ptr(), 2, STd.data_ptr(), PSd.data_ptr(), 0, None))
    torch.cuda.synchronize()
    assert abi.relerr(STd.cpu().numpy().view(np.complex128), ST) < 1e-14
    assert abi.relerr(PSd.cpu().numpy().view(np.complex128), PS) < 1e-14
    for (K, M, wu, unb) in [(5, 5, 2.0, 0), (5, 5, 1.0, 0), (5, 5, 1.5, 0), (5, 5, 2.0, 1), (1, 7, 2.0, 1), (4, 100, 2.0, 1), (5, 5, 1.0, 1)]:
        OUT = np.zeros(f.ncoef, np.complex128)
        orc.orc_weight(OUT.ctypes.data, ST.ctypes.data, PS.ctypes.data, f.ncoef, K, M, wu, unb)
        OUTd = torch.zeros_like(STd)
        tspws.check(lib.tspws_hip_weight(pl.h, OUTd.data_ptr(), STd.data_ptr(), PSd.data_ptr(), K, M, wu, unb, None))
        torch.cuda.synchronize()
        assert abi.relerr(OUTd.cpu().numpy().view(np.complex128), OUT) < 1e-13, (K, M, wu, unb)


# ------------------------------------------------------------------- partial stacks
@pytest.mark.parametrize("mtr,N,K", [(16, 2048, 4), (37, 1501, 5), (1000, 4096, 10), (7, 64, 7), (5, 1028, 1), (64, 16501, 10), (3, 8192, 3)])
def test_partial_stacks_bit_exact_groups(lib, torch, mtr, N, K):
    """FP64 sums of float32 samples: compare with the oracle; group membership must be exact."""
    X = abi.synth_traces(mtr, N, seed=9)
    P = np.zeros((K, N))
    abi.oracle().orc_partial_stacks(P.ctypes.data, X.ctypes.data, N, mtr, K)
    p = abi.resolve(abi.default_params(Kmax=K), N)
    pl = tspws.Plan(p, N)
    Xd = torch.as_tensor(X, device="cuda")
    Pd = torch.full((K, N), np.nan, dtype=torch.float64, device="cuda")
    tspws.check(lib.tspws_hip_partial_stacks(pl.h, Xd.data_ptr(), N, mtr, 0, mtr, K, Pd.data_ptr(), N, None))
    torch.cuda.synchronize()
    got = Pd.cpu().numpy()
    assert np.max(np.abs(got - P)) <= 1e-13 * max(1.0, np.max(np.abs(P)))
    # sharded: two shards with global indices, summed, equal the unsharded result
    h = mtr // 3
    A = torch.zeros_like(Pd)
    B = torch.zeros_like(Pd)
    tspws.check(lib.tspws_hip_partial_stacks(pl.h, Xd.data_ptr(), N, h, 0, mtr, K, A.data_ptr(), N, None))
    tspws.check(lib.tspws_hip_partial_stacks(pl.h, Xd[h:].data_ptr(), N, mtr - h, h, mtr, K, B.data_ptr(), N, None))
    torch.cuda.synchronize()
    assert np.max(np.abs((A + B).cpu().numpy() - P)) <= 1e-13 * max(1.0, np.max(np.abs(P)))
    # group ranges (the multi-GPU overlap streams the groups in two halves): rows outside the range stay untouched,
    # rows inside are bit-identical to the one-shot call
    if K >= 2:
        R = torch.full((K, N), np.nan, dtype=torch.float64, device="cuda")
        g = K // 2
        tspws.check(lib.tspws_hip_partial_stacks_range(pl.h, Xd.data_ptr(), N, mtr, 0, mtr, K, 0, g, R.data_ptr(), N, None))
        torch.cuda.synchronize()
        assert torch.isnan(R[g:]).all()
        tspws.check(lib.tspws_hip_partial_stacks_range(pl.h, Xd.data_ptr(), N, mtr, 0, mtr, K, g, K, R.data_ptr(), N, None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(R.cpu().numpy(), got)


def test_prologue_kernels(lib, torch):
    for N in (2048, 2047, 16501):
        X = abi.synth_traces(9, N, seed=4) + np.float32(0.25)
        Xd = torch.as_tensor(X, device="cuda").clone()
        tspws.check(lib.tspws_hip_fold(Xd.data_ptr(), 9, N, N, None))
        F = X.copy()
        h = N // 2
        v = (F[:, :h] + F[:, ::-1][:, :h]) * np.float32(0.5)
        F[:, :h] = v
        F[:, ::-1][:, :h] = v
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Xd.cpu().numpy(), F)  # float ops: bit exact
        tspws.check(lib.tspws_hip_remove_mean(Xd.data_ptr(), 9, N, N, None))
        torch.cuda.synchronize()
        m = (F.astype(np.float64).sum(axis=1) / N).astype(np.float32)
        R = F - m[:, None]
        assert np.max(np.abs(Xd.cpu().numpy() - R)) <= 1.2e-7 * np.max(np.abs(R))  # mean may round one float ulp apart


# --------------------------------------------------------------------- whole calls
def test_tspws_main_golden(lib, golden):
    """The drop-in entry point on every golden case produced by the reference."""
    g = golden["mains"]
    for name in main_case_names(g):
        check_main(lib.tspws_main, g, name, TOL32)


def test_leap_day_jackknife_goldens(lib, golden):
    """31 December of a leap year (tm_yday == 365) lands in bin n and is never deleted (ts_pws1f_lib.c:398-401): reference goldens."""
    g = golden["extra"]
    for name in main_case_names(g):
        check_main(lib.tspws_main, g, name, TOL32)


def test_tspws_main_example_data(lib, golden):
    g = golden["example32"]
    for name in ("ex1", "ex2", "ex3", "ex_mexhat"):
        p = _case_params(g, name)
        r = abi.run_main(lib.tspws_main, p, g["traces"], dt=float(g["dt"]), beg=float(g["beg"]))
        assert r["rc"] == 0
        assert abi.relerr(r["ls"], g[f"{name}/ls"]) < TOL32, name
        assert abi.relerr(r["tsPWS"], g[f"{name}/tsPWS"]) < TOL32, name


@pytest.mark.parametrize("kw,mtr,N", [
    (dict(), 24, 4096), (dict(Kmax=10, unbiased=1), 100, 8192), (dict(type=-3, Kmax=3), 30, 4096),
    (dict(w0=2 * np.pi), 12, 32768), (dict(Kmax=10, unbiased=1), 40, 131072), (dict(lrm=1, wu=1.3), 10, 3000),
    (dict(b0=3.0), 20, 49152), (dict(b0=5.0, J=7, Kmax=4, unbiased=1), 24, 20480),   # decimations 3 * 2^j, 5 * 2^j
])
def test_tspws_main_vs_oracle(lib, kw, mtr, N):
    X = abi.synth_traces(mtr, N, seed=33)
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["rc"] == 0 and b["rc"] == 0
    assert abi.relerr(a["ls"], b["ls"]) < TOL32 < NORTH_STAR_TOL
    assert abi.relerr(a["tsPWS"], b["tsPWS"]) < TOL32


def test_tspws_main_vs_reference_if_built(lib):
    ref = abi.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built on this machine")
    X = abi.synth_traces(50, 8192, seed=8)
    for kw in (dict(), dict(Kmax=10, unbiased=1), dict(type=-3)):
        p = abi.default_params(**kw)
        a = abi.run_main(lib.tspws_main, p, X)
        b = abi.run_main(ref.tspws_main, p, X)
        assert abi.relerr(a["ls"], b["ls"]) < TOL32 and abi.relerr(a["tsPWS"], b["tsPWS"]) < TOL32


def test_jackknife_vs_oracle_many_traces(lib):
    mtr, N = 400, 2048
    X = abi.synth_traces(mtr, N, seed=6)
    rng = np.random.default_rng(3)
    times = 1262304000 + 86400 * np.sort(rng.integers(0, 3 * 365, mtr))
    for kw in (dict(Kmax=10, jackknife_n=10, jackknife_d=1), dict(Kmax=4, unbiased=1, jackknife_n=6, jackknife_d=2, type=-3)):
        p = abi.default_params(**kw)
        a = abi.run_main(lib.tspws_main, p, X, times=times)
        b = abi.run_main(abi.oracle().orc_tspws_main, p, X, times=times)
        np.testing.assert_array_equal(a["jk_mtr"], b["jk_mtr"])
        for c in range(len(a["jk_mtr"])):
            assert abi.relerr(a["jk_ls"][c], b["jk_ls"][c]) < TOL32
            assert abi.relerr(a["jk_ts"][c], b["jk_ts"][c]) < TOL32
    # unsorted start times: deletion classes are not contiguous in trace order
    times2 = rng.permutation(times)
    p = abi.default_params(Kmax=5, jackknife_n=5, jackknife_d=1)
    a = abi.run_main(lib.tspws_main, p, X[:120], times=times2[:120])
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X[:120], times=times2[:120])
    np.testing.assert_array_equal(a["jk_mtr"], b["jk_mtr"])
    assert max(abi.relerr(a["jk_ts"][c], b["jk_ts"][c]) for c in range(5)) < TOL32
    # no start times: replicas are left untouched, main outputs still produced
    z = abi.run_main(lib.tspws_main, p, X[:20], times=np.zeros(20, np.int64))
    assert z["rc"] == 0 and not z["jk_ts"].any() and z["tsPWS"].any()


@pytest.mark.parametrize("env", [dict(), dict(TSPWS_JK_DIRECT="0"), dict(TSPWS_JK_STAGES="1"), dict(TSPWS_JK_STAGES="5"), dict(TSPWS_JK_PIPELINE="0"), dict(TSPWS_JK_FINAL="0")])
def test_masked_replica_engines_agree(env):
    """The streaming side of the masked replicas has three forms: rows straight from the walk (a running sum per column, <= 16
    columns), running sums with snapshots + signed sums of snapshots (any number of columns), and the class sums of the serial
    call (TSPWS_JK_PIPELINE=0); the transforms run in 1 .. Kmax stages.  Every combination against the oracle's tspws_main:
    10 replicas + the plain stack (direct walk by default), 21 replicas (n = 7, d = 2: 22 columns, the snapshot form), a replica
    that loses a whole group's worth of traces, two-stage random subsampling (masks from rand(): same engine).  Each case in a
    fresh process: the switches are read once per process."""
    import subprocess
    import sys as _sys
    e = dict(os.environ)
    e.update(env)
    e["TSPWS_LIB_PATH"] = SWEEPS_LIB   # (the shipped library does not read the switches)
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "masked_engines.py")], capture_output=True, text=True,
                       timeout=900, env=e)
    line = [l for l in r.stdout.splitlines() if l.startswith("MASKED_ENGINES")]
    assert line, r.stdout[-3000:] + r.stderr[-3000:]
    assert float(line[0].split()[1]) < TOL32, line[0]


@pytest.mark.parametrize("env", [dict(), dict(TSPWS_INV_SPLIT="0"), dict(TSPWS_INV_SPLIT="1"), dict(TSPWS_FUSE_WGS="1"), dict(TSPWS_FUSE_WGS="2048"), dict(TSPWS_FUSE_WGS="2048", TSPWS_FUSE_MINTPS="1"),
                                 dict(TSPWS_FWD_STEPS="96"), dict(TSPWS_FWD_STEPS="8"), dict(TSPWS_TL_PICK="0"), dict(TSPWS_TL_PICK="1", TSPWS_TL_MINNS1="65")])
def test_short_frame_and_many_trace_forms_agree(env):
    """Round-4 rules that pick a launch geometry by frame length / batch size -- inverse items per octave or per scale, slice
    length of the fused forward launch, tap steps per wave of the direct kernel, decomposition of the many-trace path -- each
    forced both ways: whole tspws_main calls on short frames and small ensembles against the oracle.  Fresh process per case
    (the switches are read once per process or per frame)."""
    import subprocess
    import sys as _sys
    e = dict(os.environ)
    e.update(env)
    e["TSPWS_LIB_PATH"] = SWEEPS_LIB   # (the shipped library does not read the switches)
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shape_engines.py")], capture_output=True, text=True,
                       timeout=900, env=e)
    line = [l for l in r.stdout.splitlines() if l.startswith("SHAPE_ENGINES")]
    assert line, r.stdout[-3000:] + r.stderr[-3000:]
    assert float(line[0].split()[1]) < TOL32, line[0]


def test_upload_in_pieces_of_an_unaligned_host_buffer(lib, torch):
    """tspws_hip_upload pins and copies large host buffers in page-aligned 128-MB pieces: a source that starts in the middle of a
    page and ends in the middle of a piece arrives byte for byte (and the plain small copy too)."""
    import ctypes as C
    for nbytes in (300 * (1 << 20) + 12345, 5 * (1 << 20) + 3):
        raw = np.random.default_rng(4).integers(0, 256, nbytes + 4096 + 64, dtype=np.uint8)
        off = (-raw.ctypes.data) % 4096 + 20                 # 20 bytes into a page
        src = raw[off:off + nbytes]
        dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        assert lib.tspws_hip_upload(C.c_void_p(dst.data_ptr()), C.c_void_p(src.ctypes.data), C.c_size_t(nbytes), None) == 0
        assert np.array_equal(dst.cpu().numpy(), src)
        again = np.empty(nbytes, np.uint8)
        assert lib.tspws_hip_download(C.c_void_p(again.ctypes.data), C.c_void_p(dst.data_ptr()), C.c_size_t(nbytes), None) == 0
        assert np.array_equal(again, src)


# ------------------------------------------------------- device-resident / sharded path
@pytest.mark.parametrize("kw", [dict(Kmax=10, unbiased=1), dict(), dict(type=-3, Kmax=4)])
def test_device_path_and_shards(lib, torch, kw):
    mtr, N = 64, 8192
    p = tspws.resolve(abi.default_params(**kw), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=5)
    X = Xd.cpu().numpy()
    np.testing.assert_array_equal(X, abi.synth_traces(mtr, N, seed=5) if False else X)  # (device generator is its own source)
    b = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X)
    ls, ts = pl.stack(Xd)
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), b["ls"]) < TOL32
    assert abi.relerr(ts.cpu().numpy(), b["tsPWS"]) < TOL32
    # pipelined single-GPU entry (tspws_hip_stack) gives the same answer, call after call
    for _ in range(3):
        ls1, ts1 = pl.stack_single(Xd)
    torch.cuda.synchronize()
    assert abi.relerr(ls1.cpu().numpy(), b["ls"]) < TOL32
    assert abi.relerr(ts1.cpu().numpy(), b["tsPWS"]) < TOL32
    np.testing.assert_array_equal(ts1.cpu().numpy(), ts.cpu().numpy())  # same kernels, same order
    # emulate 4 ranks on one GPU: shard-local halves, summed reduce buffers, one finish
    total = None
    for r in range(4):
        f, c = tspws.shard_range(mtr, r, 4)
        pl.stack_local(Xd[f:f + c], f, mtr)
        buf = pl.reduce_buffer(mtr).clone()
        total = buf if total is None else total + buf
    pl.reduce_buffer(mtr).copy_(total)
    ls2 = torch.empty_like(ls)
    ts2 = torch.empty_like(ts)
    pl.stack_finish(mtr, ls2, ts2)
    torch.cuda.synchronize()
    assert abi.relerr(ls2.cpu().numpy(), b["ls"]) < TOL32
    assert abi.relerr(ts2.cpu().numpy(), b["tsPWS"]) < TOL32


def test_synth_generator_matches_host_recipe(lib, torch):
    Xd = tspws.synth(5, 4096, seed=7, first=3).cpu().numpy()
    Xh = abi.synth_traces(5, 4096, seed=7, first=3)
    assert np.max(np.abs(Xd - Xh)) <= 2e-7  # same integers; sin/exp may differ by an ulp before the float cast


def test_full_size_properties(lib, torch):
    """BASELINE configs[2] at its full size (10 000 x 131 072, two-stage K = 10, unbiased): properties that do not need
    the oracle -- the group sums add up to the total, the call is linear in the data (ls and tsPWS scale with a
    positive factor), shards of the traces reduce to the unsharded buffer, and a coherent ensemble stacks to itself."""
    mtr, N, K = 10000, 131072, 10
    p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=1)
    pl.stack_local(Xd, 0, mtr)
    P = pl.reduce_buffer(mtr).view(K, N).clone()
    tot = torch.zeros(N, dtype=torch.float64, device="cuda")
    for t0 in range(0, mtr, 500):
        tot += Xd[t0:t0 + 500].double().sum(dim=0)
    assert float((P.sum(dim=0) - tot).abs().max()) <= 1e-8
    # group membership g = floor(i K / mtr): group 3 alone
    g3 = Xd[3000:4000].double().sum(dim=0)
    assert float((P[3] - g3).abs().max()) <= 1e-9
    # four shards with global indices, reduce buffers added
    acc = torch.zeros_like(P)
    for r in range(4):
        f, c = tspws.shard_range(mtr, r, 4)
        pl.stack_local(Xd[f:f + c], f, mtr)
        acc += pl.reduce_buffer(mtr).view(K, N)
    assert float((acc - P).abs().max()) <= 1e-9
    ls, ts = pl.stack(Xd)
    ls, ts = ls.clone(), ts.clone()
    # linearity: a power-of-two factor scales every intermediate exactly
    Xd *= 4.0
    ls4, ts4 = pl.stack(Xd)
    torch.cuda.synchronize()
    np.testing.assert_array_equal((ls4 / 4).cpu().numpy(), ls.cpu().numpy())
    np.testing.assert_array_equal((ts4 / 4).cpu().numpy(), ts.cpu().numpy())
    # identical traces: phase stack is fully coherent -> tsPWS == ls == ICWT(CWT(x)) (unbiased weight = 1)
    Xc = Xd[:1].repeat(64, 1).contiguous()
    lsc, tsc = pl.stack(Xc)
    torch.cuda.synchronize()
    assert abi.relerr(tsc.cpu().numpy(), lsc.cpu().numpy()) < 1e-6


def test_full_size_jackknife_properties(lib, torch):
    """BASELINE configs[3] at its full size (10 000 x 131 072, Mexican hat, two-stage K = 10, jackknife n = 10, d = 1: ten
    replicas; ts_pws1f_lib.c:719-831) through the one-pass call: the deletion plan and the replica sizes against the oracle's
    plan, every replica's K partial-stack rows against direct FP64 sums of the selected traces (group of the k-th selected
    trace: floor(k K / K_c), :766), two replicas against the plain stack of the gathered subset, the time-domain linear
    stacks, and exact scaling with a power of two."""
    mtr, N, K, n, Cn = 10000, 131072, 10, 10, 10
    p = tspws.resolve(abi.default_params(type=-3, Kmax=K, jackknife_n=n, jackknife_d=1), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=1)
    times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)   # 2010-01-01 + i days
    sel = np.zeros((Cn, mtr), np.int8)
    sel_o = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, n, Cn) == 0
    assert abi.oracle().orc_jackknife_plan(sel_o.ctypes.data, times.ctypes.data, mtr, 1, n, Cn) == 0
    np.testing.assert_array_equal(sel, sel_o)
    ls, ts, jl, jt, jm = pl.stack_jackknife(Xd, sel)
    torch.cuda.synchronize()
    ls, ts, jl, jt = ls.clone(), ts.clone(), jl.clone(), jt.clone()
    np.testing.assert_array_equal(jm, sel.sum(axis=1).astype(np.uint32))
    assert 0 < jm.min() and jm.max() < mtr
    # the stack itself: same as the plain call up to the rounding of class sums vs chunk sums
    ls0, ts0 = pl.stack(Xd)
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), ls0.cpu().numpy()) < 1e-6 and abi.relerr(ts.cpu().numpy(), ts0.cpu().numpy()) < 1e-6
    # rows of every replica (one pass over the "shard" that is the whole ensemble) against direct sums
    pl.jackknife_local(Xd, 0, mtr, sel)
    torch.cuda.synchronize()
    rows = pl.jackknife_buffer(Cn).view(Cn, K, N)
    for c in range(Cn):
        idx = np.flatnonzero(sel[c] == 1)
        Kc = idx.size
        grp = np.floor(np.arange(Kc) * K / Kc).astype(np.int64)
        for g in range(K):
            want = Xd[torch.as_tensor(idx[grp == g], device="cuda")].double().sum(dim=0)
            assert float((rows[c, g] - want).abs().max()) <= 1e-9, (c, g)
        # time-domain linear stack of the replica (:799-811): (sum of its rows) * (1 / K_c), cast to float
        lin = (rows[c].sum(dim=0) * (1.0 / Kc)).float()
        assert float((jl[c] - lin).abs().max()) <= 1e-6 * float(lin.abs().max())
    # a replica IS the two-stage stack of its selected traces
    for c in (0, 7):
        idx = torch.as_tensor(np.flatnonzero(sel[c] == 1), device="cuda")
        Xc = Xd[idx].contiguous()
        lc, tc = pl.stack(Xc)
        torch.cuda.synchronize()
        assert abi.relerr(jt[c].cpu().numpy(), tc.cpu().numpy()) < 1e-6, c
        del Xc
    # linearity: a power-of-two factor scales every intermediate exactly
    Xd *= 0.25
    ls4, ts4, jl4, jt4, jm4 = pl.stack_jackknife(Xd, sel)
    torch.cuda.synchronize()
    np.testing.assert_array_equal((ls4 * 4).cpu().numpy(), ls.cpu().numpy())
    np.testing.assert_array_equal((ts4 * 4).cpu().numpy(), ts.cpu().numpy())
    np.testing.assert_array_equal((jt4 * 4).cpu().numpy(), jt.cpu().numpy())
    np.testing.assert_array_equal((jl4 * 4).cpu().numpy(), jl.cpu().numpy())
    np.testing.assert_array_equal(jm4, jm)


def test_full_size_single_stage_properties(lib, torch):
    """BASELINE configs[1] at its full size (1024 x 32768, w0 = 2 pi, single stage): a two-stage call with one trace per
    group is the same computation, and the result is invariant under a permutation of the traces (up to summation order)."""
    mtr, N = 1024, 32768
    p1 = tspws.resolve(abi.default_params(w0=2 * np.pi), N)
    pl1 = tspws.Plan(p1, N)
    Xd = tspws.synth(mtr, N, seed=2)
    ls1, ts1 = pl1.stack(Xd)
    ls1, ts1 = ls1.clone(), ts1.clone()
    pk = tspws.resolve(abi.default_params(w0=2 * np.pi, Kmax=mtr), N)
    plk = tspws.Plan(pk, N)
    lsk, tsk = plk.stack(Xd)
    torch.cuda.synchronize()
    assert abi.relerr(tsk.cpu().numpy(), ts1.cpu().numpy()) < 1e-6
    assert abi.relerr(lsk.cpu().numpy(), ls1.cpu().numpy()) < 1e-6
    perm = torch.randperm(mtr, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    lsp, tsp = pl1.stack(Xd[perm].contiguous())
    torch.cuda.synchronize()
    assert abi.relerr(tsp.cpu().numpy(), ts1.cpu().numpy()) < 1e-6
    assert abi.relerr(lsp.cpu().numpy(), ls1.cpu().numpy()) < 1e-6


# --------------------------------------------------------- convergence / random subsampling
@pytest.mark.parametrize("kw", [dict(convergence=1, AllSteps=1), dict(convergence=1, Kmax=6, unbiased=1), dict(convergence=1, type=-3, wu=1.0)])
def test_convergence_vs_oracle(lib, kw):
    X = abi.synth_traces(40, 4096, seed=12)
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["rc"] == 0
    for k in ("conv_ls_sim", "conv_tsPWS_sim"):
        assert np.max(np.abs(a[k] - b[k])) < 1e-9, k
    # the last ts-PWS step is the full stack: similarity with its own float-rounded copy ~ 1 (the linear curve compares
    # the time-domain mean with the frame-filtered ls, so it does not reach 1)
    assert abs(a["conv_tsPWS_sim"][-1] - 1.0) < 1e-6
    for k in ("conv_ls_misfit", "conv_tsPWS_misfit"):
        assert np.max(np.abs(a[k] - b[k])) <= 1e-7 * np.max(np.abs(b[k])) + 1e-18, k
    if "conv_ts_steps" in a:
        assert abi.relerr(a["conv_ts_steps"], b["conv_ts_steps"]) < TOL32
        assert abi.relerr(a["conv_ls_steps"], b["conv_ls_steps"]) < TOL32
        np.testing.assert_array_equal(a["conv_ts_steps"][-1], a["tsPWS"])


def test_concurrent_callers_of_the_drop_in(lib):
    """Two host threads call tspws_main at the same time with different ensembles and frames (the reference has no shared state;
    this engine keeps a cached frame and trace buffer, so the calls are serialised inside): both get their own results."""
    import threading
    cases = [(dict(), abi.synth_traces(20, 4096, seed=41)), (dict(type=-3, Kmax=4, unbiased=1), abi.synth_traces(33, 3001, seed=42)),
             (dict(w0=2 * np.pi), abi.synth_traces(12, 8192, seed=43))]
    want = [abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X) for kw, X in cases]
    got = [[None] * 4 for _ in cases]

    def work(i):
        for r in range(4):
            got[i][r] = abi.run_main(lib.tspws_main, abi.default_params(**cases[i][0]), cases[i][1])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, w in enumerate(want):
        for r in range(4):
            a = got[i][r]
            assert a is not None and a["rc"] == 0
            assert abi.relerr(a["ls"], w["ls"]) < TOL32 and abi.relerr(a["tsPWS"], w["tsPWS"]) < TOL32, (i, r)


def test_per_device_slots_and_the_device_list_run_side_by_side(lib, monkeypatch):
    """The drop-in's cache is a table with one slot and one lock per device plus one entry for the several-device call.  On a one-GPU
    box two different cache entries of device 0 can be busy at once: thread A stacks through tspws_main_on(0, ...) (the device-0
    slot), thread B through tspws_main under TSPWS_DEVICES=0,0 (the several-device entry: two virtual shards) -- concurrently, each
    with its own frame, both right.  Afterwards the device-0 slot holds A's frame; tspws_main_release empties the table."""
    import threading
    lib.tspws_main_release()
    assert lib.tspws_main_cached_devices() == 0
    XA, XB = abi.synth_traces(24, 4096, seed=51), abi.synth_traces(31, 2048, seed=52)
    kwA, kwB = dict(Kmax=6, unbiased=1), dict(type=-3, Kmax=4)
    wantA = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kwA), XA)
    wantB = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kwB), XB)
    monkeypatch.setenv("TSPWS_DEVICES", "0,0")
    got = {"A": [], "B": []}

    def A():
        for _ in range(6):
            got["A"].append(abi.run_main(lambda p, o, d: lib.tspws_main_on(0, p, o, d), abi.default_params(**kwA), XA))

    def B():
        for _ in range(6):
            got["B"].append(abi.run_main(lib.tspws_main, abi.default_params(**kwB), XB))

    th = [threading.Thread(target=A), threading.Thread(target=B)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    try:
        for r in got["A"]:
            assert r["rc"] == 0 and abi.relerr(r["tsPWS"], wantA["tsPWS"]) < TOL32 and abi.relerr(r["ls"], wantA["ls"]) < TOL32
        for r in got["B"]:
            assert r["rc"] == 0 and abi.relerr(r["tsPWS"], wantB["tsPWS"]) < TOL32 and abi.relerr(r["ls"], wantB["ls"]) < TOL32
        assert lib.tspws_main_cached_devices() == 1          # the device-0 slot (A's frame); B used the several-device entry
        # the environment names the device of tspws_main; tspws_main_on ignores it
        monkeypatch.setenv("TSPWS_DEVICES", "0")
        r = abi.run_main(lib.tspws_main, abi.default_params(**kwA), XA)
        assert r["rc"] == 0 and abi.relerr(r["tsPWS"], wantA["tsPWS"]) < TOL32
        monkeypatch.setenv("TSPWS_DEVICES", "7")            # no such device on this box
        assert abi.run_main(lib.tspws_main, abi.default_params(**kwA), XA)["rc"] == 5
        assert abi.run_main(lambda p, o, d: lib.tspws_main_on(0, p, o, d), abi.default_params(**kwA), XA)["rc"] == 0
    finally:
        lib.tspws_main_release()
    assert lib.tspws_main_cached_devices() == 0


def test_local_backend_is_refused_for_distinct_devices(lib, monkeypatch):
    """TSPWS_COMM=local is the one-GPU test vehicle: with distinct devices in the list it must not stand in for RCCL.  A one-GPU box
    can only show the refusal of a list that is not all one device through the argument check (device 1 does not exist here: 5),
    and that repeated-device lists still select it."""
    h = C.c_void_p()
    arr = (C.c_int * 2)(0, 0)
    assert lib.tspws_hip_comm_create(C.byref(h), 2, arr) == 0
    assert lib.tspws_hip_comm_backend(h).decode() == "local"
    lib.tspws_hip_comm_destroy(h)
    if lib.tspws_hip_device_count() >= 2:
        monkeypatch.setenv("TSPWS_COMM", "local")
        arr = (C.c_int * 2)(0, 1)
        assert lib.tspws_hip_comm_create(C.byref(h), 2, arr) == -1
        assert b"local backend" in lib.tspws_hip_last_error()


def test_seeded_subsampling_in_a_fresh_process():
    """srand(seed) followed by a process's FIRST tspws_main call: the random subsamples are the reference's (the masks are drawn
    before the HIP runtime initialises, which consumes rand() values)."""
    import subprocess
    import sys as _sys
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fresh_subsample.py")],
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("FRESH_SUBSAMPLE")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    _, rc_a, rc_b, worst = line[0].split()
    assert rc_a == "0" and rc_b == "0" and float(worst) < TOL32, line[0]


@pytest.mark.parametrize("kw", [dict(subsmpl_N=4, subsmpl_p=0.3), dict(subsmpl_N=3, subsmpl_p=0.7, Kmax=5, unbiased=1),
                                dict(subsmpl_N=2, subsmpl_p=1.0, type=-3)])
def test_random_subsampling_vs_oracle(lib, kw):
    X = abi.synth_traces(50, 4096, seed=13)
    p = abi.default_params(**kw)
    abi.srand(7)
    a = abi.run_main(lib.tspws_main, p, X)
    abi.srand(7)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["rc"] == 0
    for m in range(kw["subsmpl_N"]):
        assert abi.relerr(a["sub_ls"][m], b["sub_ls"][m]) < TOL32, m
        assert abi.relerr(a["sub_ts"][m], b["sub_ts"][m]) < TOL32, m
    if kw["subsmpl_p"] == 1.0 and not kw.get("Kmax"):  # every trace kept: each subsample is the full single-stage stack
        assert abi.relerr(a["sub_ts"][0], a["tsPWS"]) < TOL32


# ------------------------------------------------------------------------------- layout edge cases
@pytest.mark.parametrize("pad", [4, 1])
def test_padded_rows_and_many_traces(sweeps, torch, pad, monkeypatch):
    """Row stride ld > N (vectorised when ld % 4 == 0, scalar otherwise) through the device-resident path."""
    sw, swlib = sweeps   # the build that reads TSPWS_TL_MIN / TSPWS_TL_BATCH (conftest.py)
    monkeypatch.setenv("TSPWS_TL_MIN", "64")   # (the single-stage part below goes through the many-trace kernel)
    mtr, N, K = 70, 4096, 7
    p = sw.resolve(abi.default_params(Kmax=K, unbiased=1), N)
    pl = sw.Plan(p, N)
    X = abi.synth_traces(mtr, N, seed=44)
    buf = torch.zeros((mtr, N + pad), dtype=torch.float32, device="cuda")
    buf[:, :N] = torch.as_tensor(X, device="cuda")
    view = buf[:, :N]  # shape (mtr, N), stride (N + pad, 1)
    assert view.stride(0) == N + pad
    ls, ts = pl.stack(view)
    torch.cuda.synchronize()
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(Kmax=K, unbiased=1), X)
    assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32
    # single-stage through the same strided view (float input path of the forward kernels)
    p1 = sw.resolve(abi.default_params(), N)
    pl1 = sw.Plan(p1, N)
    ls1, ts1 = pl1.stack(view[:12])
    torch.cuda.synchronize()
    want1 = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(), X[:12])
    assert abi.relerr(ts1.cpu().numpy(), want1["tsPWS"]) < TOL32 and abi.relerr(ls1.cpu().numpy(), want1["ls"]) < TOL32


@pytest.mark.parametrize("pad", [4, 3])
def test_shipped_library_batches_and_padded_rows(lib, torch, pad):
    """The SHIPPED library's own thresholds (the path-forcing tests above run on the -DTSPWS_SWEEPS build): a single-stage ensemble of 5000
    traces x 1024 samples is walked in two batches of the many-trace path (4096 + 904 traces, the second batch partially filled) with the
    default engine rule (>= 256 traces: spectral chain beside the trace-lane kernel); the same through rows that are ld = N + pad samples apart,
    and a 300 x 3000 call (N not a power of two: window of the periodic extension) through padded rows."""
    mtr, N = 5000, 1024
    X = abi.synth_traces(mtr, N, seed=52)
    X[4500] = 0
    p = abi.default_params(wu=1.0)
    want = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    pl = tspws.Plan(tspws.resolve(p, N), N)
    assert lib.tspws_hip_spectral_choice(pl.h, mtr) < pl.S
    buf = torch.zeros((mtr, N + pad), dtype=torch.float32, device="cuda")
    buf[:, :N] = torch.as_tensor(X, device="cuda")
    for view in (torch.as_tensor(X, device="cuda"), buf[:, :N]):
        ls, ts = pl.stack(view)
        torch.cuda.synchronize()
        assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32, view.stride(0)
    mtr, N = 300, 3000
    X = abi.synth_traces(mtr, N, seed=53)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(), X)
    pl = tspws.Plan(tspws.resolve(abi.default_params(), N), N)
    buf = torch.zeros((mtr, N + pad), dtype=torch.float32, device="cuda")
    buf[:, :N] = torch.as_tensor(X, device="cuda")
    ls, ts = pl.stack(buf[:, :N])
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32


def test_plan_reuse_and_argument_errors(lib, torch):
    N = 2048
    p = tspws.resolve(abi.default_params(Kmax=4), N)
    pl = tspws.Plan(p, N)
    X = tspws.synth(16, N, seed=9)
    a = [t.cpu().numpy() for t in pl.stack(X)]
    b = [t.cpu().numpy() for t in pl.stack(X)]       # same plan, same inputs: bit-identical (deterministic reductions)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    h = C.c_void_p()
    assert lib.tspws_hip_plan_create(C.byref(h), 1, 4, 4, N, 2.0, 1.0, abi.W0_DEFAULT, 0, 0) == 7      # real families are out of scope
    assert lib.tspws_hip_plan_create(C.byref(h), -1, 0, 4, N, 2.0, 1.0, abi.W0_DEFAULT, 0, 0) == 7     # empty frame
    assert lib.tspws_hip_plan_create(C.byref(h), -1, 4, 4, N, 2.0, 1.0, abi.W0_DEFAULT, 0, 99) == 5    # no such device
    assert lib.tspws_hip_partial_stacks(pl.h, None, N, 4, 0, 4, 2, None, N, None) == -1
    assert b"partial_stacks" in lib.tspws_hip_last_error()


@pytest.mark.gpu
def test_more_than_2_to_32_samples(lib, torch):
    """BASELINE configs[4] shards 100 000 x 131 072 traces over 8 GPUs; the reference cannot index that (32-bit itr*max,
    SURVEY section 8).  One GPU with 40 000 traces = 5.2e9 samples crosses 2^32 element offsets: group sums and the
    last trace must still land where they belong."""
    mtr, N, K = 40000, 131072, 10
    p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=3)
    assert Xd.numel() > 2 ** 32
    pl.stack_local(Xd, 0, mtr)
    P = pl.reduce_buffer(mtr).view(K, N).clone()
    g9 = torch.zeros(N, dtype=torch.float64, device="cuda")
    for t0 in range(36000, 40000, 500):
        g9 += Xd[t0:t0 + 500].double().sum(dim=0)
    assert float((P[9] - g9).abs().max()) <= 1e-9
    before = Xd[-1].double()
    Xd[-1] += 1.0  # only the very last trace changes -> only group 9 moves, by the (float-rounded) increment
    delta = Xd[-1].double() - before
    pl.stack_local(Xd, 0, mtr)
    P2 = pl.reduce_buffer(mtr).view(K, N)
    assert float((P2[:9] - P[:9]).abs().max()) == 0.0
    assert float((P2[9] - P[9] - delta).abs().max()) <= 1e-9
    ls, ts = pl.stack(Xd)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ts).all()) and bool(torch.isfinite(ls).all())


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(Kmax=10, unbiased=1), dict(Kmax=7), dict(type=-3, Kmax=4, wu=1.0)])
def test_finish_in_pieces_is_bit_identical(lib, torch, kw):
    """The multi-GPU orchestration transforms the first half of the groups while the second half is still being reduced:
    stack_finish_range(0, h) + stack_finish_range(h, K) + stack_finish_tail must equal stack_finish bit for bit."""
    mtr, N = 64, 8192
    p = tspws.resolve(abi.default_params(**kw), N)
    pl = tspws.Plan(p, N)
    K = p.Kmax
    Xd = tspws.synth(mtr, N, seed=6)
    ls0, ts0 = pl.stack(Xd)
    ls0, ts0 = ls0.clone(), ts0.clone()
    pl.stack_local(Xd, 0, mtr)
    ls1 = torch.empty_like(ls0)
    ts1 = torch.empty_like(ts0)
    h = K // 2
    pl.stack_finish_range(mtr, 0, h)
    pl.stack_finish_range(mtr, h, K)
    pl.stack_finish_tail(mtr, ls1, ts1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ls1.cpu().numpy(), ls0.cpu().numpy())
    np.testing.assert_array_equal(ts1.cpu().numpy(), ts0.cpu().numpy())
    with pytest.raises(tspws.TspwsError):
        pl.stack_finish_range(mtr, 3, K + 1)


@pytest.mark.gpu
def test_empty_shard_and_wrapper_checks(lib, torch):
    """A rank whose shard is empty (mtr_global < world) must still reach the collective: partial_stacks_range and
    stack_local accept a (0, N) tensor (data_ptr() == 0) and leave zero rows.  The ctypes wrappers reject tensors the C ABI
    would silently reinterpret."""
    N, K, mtr_global = 2048, 2, 2
    p = tspws.resolve(abi.default_params(Kmax=K), N)
    pl = tspws.Plan(p, N)
    X = tspws.synth(mtr_global, N, seed=11)
    buf = pl.reduce_buffer(mtr_global)
    buf.fill_(7.0)
    pl.partial_stacks_range(X[:0], 0, mtr_global, 0, 1)
    pl.partial_stacks_range(X[:0], 0, mtr_global, 1, K)
    torch.cuda.synchronize()
    assert float(buf.abs().max()) == 0.0
    # three "ranks" (empty, trace 0, trace 1) sum to the unsharded partial stacks
    total = torch.zeros_like(buf)
    for f, c in ((0, 0), (0, 1), (1, 1)):
        pl.partial_stacks_range(X[f:f + c], f, mtr_global, 0, K)
        total += pl.reduce_buffer(mtr_global)
    pl.stack_local(X, 0, mtr_global)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(total.cpu().numpy(), pl.reduce_buffer(mtr_global).cpu().numpy())
    # shard outside the ensemble
    assert lib.tspws_hip_partial_stacks(pl.h, X.data_ptr(), N, 2, 1, 2, K, buf.data_ptr(), N, None) == -1
    # wrapper argument checks
    with pytest.raises(tspws.TspwsError):
        pl.stack_local(X.double(), 0, mtr_global)
    with pytest.raises(tspws.TspwsError):
        pl.stack_local(X.t().contiguous().t(), 0, mtr_global)     # column-major view: stride(1) != 1
    with pytest.raises(tspws.TspwsError):
        pl.stack_local(X[:, : N // 2], 0, mtr_global)             # wrong trace length
    with pytest.raises(tspws.TspwsError):
        pl.stack_single(X, ls=torch.empty(N, dtype=torch.float64, device=X.device))


@pytest.mark.gpu
def test_reduce_buffer_survives_growth(lib, torch):
    """A reduce-buffer view handed to a collective must stay valid when the same plan later needs a larger buffer
    (another Kmax / a single-stage call): the outgrown block is retired, not freed, and a fresh view is returned."""
    N = 2048
    p = tspws.resolve(abi.default_params(Kmax=2), N)
    pl = tspws.Plan(p, N)
    X = tspws.synth(8, N, seed=12)
    pl.stack_local(X, 0, 8)
    small = pl.reduce_buffer(8)
    keep = small.clone()
    pl.params.Kmax = 8                        # same plan, four times the partial-stack rows
    big = pl.reduce_buffer(8)
    assert big.numel() == 8 * N and big.data_ptr() != small.data_ptr()
    big.fill_(1.0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(small.cpu().numpy(), keep.cpu().numpy())   # old view still readable, untouched
    pl.params.Kmax = 2
    ls, ts = pl.stack(X)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(Kmax=2), X.cpu().numpy())
    assert abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32


@pytest.mark.gpu
def test_stack_and_jackknife_share_the_streaming_pass(lib, torch):
    """tspws_hip_stack_jackknife: ONE pass over the traces for the stack's own groups and for every replica.  Same replicas
    as the stand-alone jackknife (which streams by itself), same main outputs as the plain call (class sums instead of chunk
    sums: rounding only), and everything against the oracle's tspws_main; single-stage parameters fall back to the plain stack."""
    mtr, N, K, n = 300, 4096, 6, 5
    p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1, jackknife_n=n, jackknife_d=1), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=13)
    rng = np.random.default_rng(5)
    times = (1262304000 + 86400 * np.sort(rng.integers(0, 2 * 365, mtr))).astype(np.int64)
    Cn = n
    sel = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, n, Cn) == 0

    def jack(selection):
        jl = torch.empty((Cn, N), dtype=torch.float32, device="cuda")
        jt = torch.empty((Cn, N), dtype=torch.float32, device="cuda")
        jm = np.zeros(Cn, np.uint32)
        tspws.check(lib.tspws_hip_jackknife(pl.h, C.byref(pl.params), Xd.data_ptr(), N, mtr, selection.ctypes.data, Cn, jl.data_ptr(), jt.data_ptr(),
                                            jm.ctypes.data, None), "jackknife")
        torch.cuda.synchronize()
        return jl.cpu().numpy(), jt.cpu().numpy(), jm

    ls0, ts0 = [t.cpu().numpy() for t in pl.stack(Xd)]       # plain call, stand-alone jackknife
    l0, t0, m0 = jack(sel)
    ls1, ts1, l1, t1, m1 = pl.stack_jackknife(Xd, sel)       # one pass for groups + replicas
    torch.cuda.synchronize()
    ls1, ts1, l1, t1 = [t.cpu().numpy() for t in (ls1, ts1, l1, t1)]
    np.testing.assert_array_equal(m0, m1)
    assert abi.relerr(ls1, ls0) < 1e-6 and abi.relerr(ts1, ts0) < 1e-6   # class sums instead of chunk sums: rounding only
    assert abi.relerr(l1, l0) < 1e-6 and abi.relerr(t1, t0) < 1e-6       # (the plain groups split some classes: rounding only)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(Kmax=K, unbiased=1, jackknife_n=n, jackknife_d=1), Xd.cpu().numpy(), times=times)
    assert abi.relerr(ts1, want["tsPWS"]) < TOL32 and abi.relerr(ls1, want["ls"]) < TOL32
    assert max(abi.relerr(t1[c], want["jk_ts"][c]) for c in range(Cn)) < TOL32 and max(abi.relerr(l1[c], want["jk_ls"][c]) for c in range(Cn)) < TOL32
    np.testing.assert_array_equal(m1, want["jk_mtr"])
    # a second call with another selection on the same plan: nothing is carried over
    sel2 = sel[::-1].copy()
    _, _, l2, t2, m2 = pl.stack_jackknife(Xd, sel2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t2.cpu().numpy(), t1[::-1])
    np.testing.assert_array_equal(m2, m1[::-1])
    # single-stage parameters: the plain stack, replica outputs untouched
    p1 = tspws.resolve(abi.default_params(), N)
    pl1 = tspws.Plan(p1, N)
    a, b, jl, jt, jm = pl1.stack_jackknife(Xd[:40], sel[:, :40])
    a0, b0 = pl1.stack(Xd[:40])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a.cpu().numpy(), a0.cpu().numpy())
    np.testing.assert_array_equal(b.cpu().numpy(), b0.cpu().numpy())
    assert not jm.any()


@pytest.mark.gpu
@pytest.mark.parametrize("kw,bounds", [
    (dict(Kmax=6, unbiased=1, jackknife_n=5, jackknife_d=1), (0, 97, 97, 211, 300)),     # ragged shards, one of them EMPTY
    (dict(Kmax=4, jackknife_n=6, jackknife_d=2, type=-3), (0, 150, 300)),
])
def test_sharded_jackknife_rows_add_up(lib, torch, kw, bounds):
    """SURVEY 8e, jackknife sharding, through the C ABI in one process: every shard walks its traces once
    (tspws_hip_jackknife_local), the shards' rows are added by hand (what the collective does), replicas are finished in two
    ranges (what two owners do) -- against the oracle's tspws_main and against the unsharded engine."""
    mtr, N = bounds[-1], 4096
    p = tspws.resolve(abi.default_params(**kw), N)
    K = p.Kmax
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=21)
    rng = np.random.default_rng(9)
    times = (1262304000 + 86400 * np.sort(rng.integers(0, 2 * 365, mtr))).astype(np.int64)
    Cn = abi.binomial(p.jackknife_n, p.jackknife_d)
    sel = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, p.jackknife_d, p.jackknife_n, Cn) == 0
    main_sum = torch.zeros(K * N, dtype=torch.float64, device="cuda")
    rows_sum = torch.zeros(Cn * K * N, dtype=torch.float64, device="cuda")
    for a, b in zip(bounds[:-1], bounds[1:]):
        pl.jackknife_local(Xd[a:b], a, mtr, sel)
        torch.cuda.synchronize()
        main_sum += pl.reduce_buffer(mtr)
        rows_sum += pl.jackknife_buffer(Cn)
    pl.reduce_buffer(mtr).copy_(main_sum)
    pl.jackknife_buffer(Cn).copy_(rows_sum)
    ls = torch.empty(N, dtype=torch.float32, device="cuda")
    ts = torch.empty(N, dtype=torch.float32, device="cuda")
    pl.stack_finish(mtr, ls, ts)
    jl = torch.zeros((Cn, N), dtype=torch.float32, device="cuda")
    jt = torch.zeros((Cn, N), dtype=torch.float32, device="cuda")
    jm = np.zeros(Cn, np.uint32)
    cut = Cn // 2 + 1
    pl.jackknife_finish(mtr, sel, cut, Cn, jl, jt, jm)   # the ranges in any order
    pl.jackknife_finish(mtr, sel, 0, cut, jl, jt, jm)
    pl.jackknife_finish(mtr, sel, 2, 2, jl, jt, jm)      # empty range: nothing happens
    torch.cuda.synchronize()
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), Xd.cpu().numpy(), times=times)
    np.testing.assert_array_equal(jm, want["jk_mtr"])
    assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32
    for c in range(Cn):
        assert abi.relerr(jl[c].cpu().numpy(), want["jk_ls"][c]) < TOL32
        assert abi.relerr(jt[c].cpu().numpy(), want["jk_ts"][c]) < TOL32
    # world 1 through the orchestration: same kernels on one shard, no collective
    l1, t1, jl1, jt1, jm1 = tspws.jackknife_sharded(pl, Xd, sel)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(jm1, want["jk_mtr"])
    assert abi.relerr(t1.cpu().numpy(), want["tsPWS"]) < TOL32
    assert max(abi.relerr(jt1[c].cpu().numpy(), jt[c].cpu().numpy()) for c in range(Cn)) < 1e-6  # shard boundaries move roundings only
    assert max(abi.relerr(jl1[c].cpu().numpy(), jl[c].cpu().numpy()) for c in range(Cn)) < 1e-6
    # argument checks
    with pytest.raises(tspws.TspwsError):
        pl.jackknife_local(Xd[:10], 295, mtr, sel)       # shard sticks out of the ensemble
    with pytest.raises(tspws.TspwsError):
        pl.jackknife_finish(mtr, sel, 3, Cn + 1, jl, jt, jm)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,N,mtr", [
    (dict(), 4096, 64), (dict(), 2048, 200), (dict(), 1501, 70), (dict(type=-3), 4096, 100), (dict(w0=2 * np.pi), 8192, 129),
    (dict(s0=3.7, J=6), 3001, 77), (dict(type=-2, wu=1.0), 2048, 90), (dict(unbiased=1), 16501, 96), (dict(uni=1, J=3), 1024, 65),
    (dict(b0=4.0), 8192, 80), (dict(b0=3.0), 12288, 70),
    (dict(Kmax=80, unbiased=1), 2048, 200),   # two-stage with many groups: the 80 FP64 partial stacks take the many-trace path too
])
def test_many_trace_single_stage_vs_oracle(sweeps, torch, kw, N, mtr, monkeypatch):
    """Single-stage stacks of >= 64 traces run on the trace-lane kernel (csrc/fwd_tl.h: transposed batch, lanes = traces, fused
    phase stack per 64-trace block, residue splits for large decimations, direct kernel for the coarsest scales): whole call
    against the oracle, device-resident and through tspws_main, including partially filled trace blocks and an all-zero trace.
    (TSPWS_TL_MIN forces that path: by default ensembles this small stay on the few-trace kernels.)"""
    sw, swlib = sweeps   # the build that reads TSPWS_TL_MIN / TSPWS_TL_BATCH (conftest.py)
    monkeypatch.setenv("TSPWS_TL_MIN", "64")
    X = abi.synth_traces(mtr, N, seed=41)
    X[mtr // 3] = 0.0
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X)
    pl = sw.Plan(sw.resolve(abi.default_params(**kw), N), N)
    ls, ts = pl.stack(torch.as_tensor(X, device="cuda"))
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32
    got = abi.run_main(swlib.tspws_main, abi.default_params(**kw), X)
    assert abi.relerr(got["ls"], want["ls"]) < TOL32 and abi.relerr(got["tsPWS"], want["tsPWS"]) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("kw,mtr,N", [(dict(), 100, 8192), (dict(), 1000, 8192), (dict(type=-3), 1000, 8192)])
def test_single_stage_path_choice_is_invisible(sweeps, torch, kw, mtr, N, monkeypatch):
    """The library picks the forward path of a single-stage batch by its size and frame (few-trace kernels below ~7 M samples and
    for two-voice frames, the trace-lane kernel above): forced either way and left alone, the call gives the oracle's outputs."""
    sw, swlib = sweeps   # the build that reads TSPWS_TL_MIN / TSPWS_TL_BATCH (conftest.py)
    X = abi.synth_traces(mtr, N, seed=48)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X)
    Xd = torch.as_tensor(X, device="cuda")
    for force in (None, "64", "1000000"):
        if force is None:
            monkeypatch.delenv("TSPWS_TL_MIN", raising=False)
        else:
            monkeypatch.setenv("TSPWS_TL_MIN", force)
        pl = sw.Plan(sw.resolve(abi.default_params(**kw), N), N)
        ls, ts = pl.stack(Xd)
        torch.cuda.synchronize()
        assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32, force


@pytest.mark.gpu
def test_many_trace_batches(sweeps, torch, monkeypatch):
    """The trace-lane path walks large ensembles in batches (transposed copy <= 1 GiB): forced here to 64 / 128 traces per batch,
    the later batches add to the stacks of the first, the last one is partial."""
    sw, swlib = sweeps   # the build that reads TSPWS_TL_MIN / TSPWS_TL_BATCH (conftest.py)
    monkeypatch.setenv("TSPWS_TL_MIN", "64")
    mtr, N = 200, 2048
    X = abi.synth_traces(mtr, N, seed=47)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(), X)
    for b in ("64", "128"):
        monkeypatch.setenv("TSPWS_TL_BATCH", b)
        pl = sw.Plan(sw.resolve(abi.default_params(), N), N)
        ls, ts = pl.stack(torch.as_tensor(X, device="cuda"))
        torch.cuda.synchronize()
        assert abi.relerr(ls.cpu().numpy(), want["ls"]) < TOL32 and abi.relerr(ts.cpu().numpy(), want["tsPWS"]) < TOL32


@pytest.mark.gpu
def test_many_trace_path_matches_the_few_trace_kernels(sweeps, torch, monkeypatch):
    """Same ensemble through the trace-lane decomposition (stack_local on 192 traces: the single-stage all-reduce payload
    ST || PS) and through the per-trace forward API (k_fwd_lds / k_fwd_poly, coefficients of every trace) with the stacks
    formed on the host: the FP64 stacks agree to rounding."""
    sw, swlib = sweeps   # the build that reads TSPWS_TL_MIN / TSPWS_TL_BATCH (conftest.py)
    monkeypatch.setenv("TSPWS_TL_MIN", "64")
    mtr, N = 192, 8192
    p = sw.resolve(abi.default_params(), N)
    Xd = sw.synth(mtr, N, seed=43)
    pl = sw.Plan(p, N)
    pl.stack_local(Xd, 0, mtr)
    torch.cuda.synchronize()
    buf = pl.reduce_buffer(mtr).cpu().numpy()
    nc = pl.ncoef
    ST = buf[:2 * nc].view(np.complex128)
    PS = buf[2 * nc:].view(np.complex128)
    Y = dev_forward(torch, pl, Xd.cpu().numpy())            # [mtr][ncoef] from the few-trace kernels
    mag = np.abs(Y)
    U = np.where(mag > 0, Y / np.where(mag > 0, mag, 1.0), 0.0)
    assert abi.relerr(ST, Y.sum(axis=0)) < TOL64
    assert abi.relerr(PS, U.sum(axis=0)) < TOL64


def _random_case(rng):
    """One random parameter set within the CLI's grammar (ts_pws1f.c:163-215): wavelet, Q / cycles / w0, fixed or automatic
    V / s0 / b0 / J, fmin, power, unbiased, two-stage, rm, fold; odd, prime and power-of-two lengths."""
    kw = {}
    typ = int(rng.choice([-1, -1, -1, -2, -3]))
    kw["type"] = typ
    if typ != -3:
        how = rng.integers(0, 4)
        if how == 1:
            kw.update(Q=float(rng.uniform(2.0, 9.0)), w0set=1)
        elif how == 2:
            kw.update(cycle=float(rng.uniform(1.0, 5.0)), w0set=2)
        elif how == 3:
            kw["w0"] = float(rng.uniform(2.5, 13.0))   # below ~3.8: b0 = 0 -> every decimation 1
    if rng.random() < 0.3:
        kw["V"] = int(rng.integers(1, 7))
    if rng.random() < 0.3:
        kw["s0"] = float(rng.uniform(1.0, 5.0))
    if rng.random() < 0.3:
        kw["b0"] = float(rng.choice([0.5, 1.0, 2.0, 3.0, 4.0]))
    if rng.random() < 0.4:
        kw["J"] = int(rng.integers(1, 7))
    elif rng.random() < 0.3:
        kw["fmin"] = float(rng.uniform(0.002, 0.05))
    kw["wu"] = float(rng.choice([2.0, 2.0, 1.0, 1.5, 0.5, 3.0]))
    if kw["wu"] == 2.0 and rng.random() < 0.5:
        kw["unbiased"] = 1
    if rng.random() < 0.5:
        kw["Kmax"] = int(rng.integers(1, 13))
    if rng.random() < 0.3:
        kw["lrm"] = 1
    N = int(rng.choice([256, 257, 509, 640, 1000, 1024, 1501, 2048, 2311, 3000, 4096, 5003]))
    mtr = int(rng.choice([1, 2, 3, 7, 16, 33, 70, 100]))
    beg = 0.0
    if rng.random() < 0.25:
        kw["fold"] = 1
        beg = -0.5 * (N - 1) if rng.random() < 0.8 else 0.0   # symmetric lag axis (fold applies) or not (warning, ignored)
    return kw, N, mtr, beg


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_parameter_sets_vs_oracle(lib, seed):
    """Seeded random sweep over the parameter grammar: return codes, resolved parameters (the call mutates *tspws like the
    reference, ts_pws1f_lib.c:91-124), mutated traces (fold / rm) and both outputs against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    for it in range(6):
        kw, N, mtr, beg = _random_case(rng)
        X = abi.synth_traces(mtr, N, seed=100 * seed + it)
        p = abi.default_params(**kw)
        a = abi.run_main(lib.tspws_main, p, X, beg=beg)
        b = abi.run_main(abi.oracle().orc_tspws_main, p, X, beg=beg)
        tag = f"seed {seed} case {it}: {kw} N={N} mtr={mtr} beg={beg}"
        assert a["rc"] == b["rc"], tag
        for f in ("J", "V", "fold"):
            assert getattr(a["params"], f) == getattr(b["params"], f), tag
        for f in ("s0", "b0", "w0"):
            assert getattr(a["params"], f) == getattr(b["params"], f), tag
        np.testing.assert_array_equal(a["sigall"], b["sigall"], err_msg=tag)   # (also when the call fails: fold / rm come first)
        if a["rc"]:
            continue
        assert abi.relerr(a["ls"], b["ls"]) < TOL32, tag
        assert abi.relerr(a["tsPWS"], b["tsPWS"]) < TOL32, tag


@pytest.mark.gpu
@pytest.mark.parametrize("kw,N,world", [
    (dict(Kmax=10, unbiased=1), 131072, 8), (dict(Kmax=5), 16501, 3), (dict(type=-3, Kmax=4, wu=1.0), 8192, 4),
    (dict(Kmax=3, w0=3.0), 4096, 2),    # b0 = 0: every decimation is 1 -> ONE octave: ranks 1.. get empty shares
    (dict(Kmax=6, unbiased=1, V=3), 3001, 5),
])
def test_scale_sharded_finish_adds_up(lib, torch, kw, N, world):
    """Sharded finish stage (multi-GPU): the ranks' shares of the scales partition the frame, every share is a run of whole
    decimation octaves, and the partial reconstructions of all shares add up to the plain finish -- same kernels on
    sub-ranges of their launch lists, only the order of the final sum over octaves differs."""
    mtr = 60
    p = tspws.resolve(abi.default_params(**kw), N)
    pl = tspws.Plan(p, N)
    Xd = tspws.synth(mtr, N, seed=31)
    ls0 = torch.empty(N, dtype=torch.float32, device="cuda")
    ts0 = torch.empty(N, dtype=torch.float32, device="cuda")
    pl.stack_local(Xd, 0, mtr)
    pl.stack_finish(mtr, ls0, ts0)
    shares = [pl.finish_shard(mtr, r, world) for r in range(world)]
    assert all(s is not None for s in shares)
    D = pl.tables()["D"]
    covered = []
    for a, b in shares:
        assert 0 <= a <= b <= pl.S
        if a < b:
            assert a == 0 or D[a - 1] != D[a]          # starts an octave
            assert b == pl.S or D[b - 1] != D[b]        # ends one
            covered += list(range(a, b))
    assert covered == list(range(pl.S))                # a partition, in rank order
    total = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
    x2 = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    for a, b in shares[::-1]:                           # any order: each call recomputes its scales from the reduce buffer
        pl.stack_finish_scales(mtr, a, b, x2)
        total += x2
    ls1 = torch.empty(N, dtype=torch.float32, device="cuda")
    ts1 = torch.empty(N, dtype=torch.float32, device="cuda")
    pl.epilogue(total, mtr, ls1, ts1)
    torch.cuda.synchronize()
    assert abi.relerr(ls1.cpu().numpy(), ls0.cpu().numpy()) < 1e-6 and abi.relerr(ts1.cpu().numpy(), ts0.cpu().numpy()) < 1e-6
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), Xd.cpu().numpy())
    assert abi.relerr(ts1.cpu().numpy(), want["tsPWS"]) < TOL32 and abi.relerr(ls1.cpu().numpy(), want["ls"]) < TOL32
    # the plain finish is untouched by the range state
    ls2 = torch.empty(N, dtype=torch.float32, device="cuda")
    ts2 = torch.empty(N, dtype=torch.float32, device="cuda")
    pl.stack_finish(mtr, ls2, ts2)
    torch.cuda.synchronize()
    assert torch.equal(ls2, ls0) and torch.equal(ts2, ts0)
    with pytest.raises(tspws.TspwsError):
        pl.stack_finish_scales(mtr, 1, pl.S, x2)        # not an octave boundary (V >= 2 everywhere here) ... or a single octave
    # single-stage parameters have no sharded finish
    pl1 = tspws.Plan(tspws.resolve(abi.default_params(), N), N)
    assert pl1.finish_shard(mtr, 0, 2) is None


def _gpu_shard_worker(rank, world, port, kw, mtr, N, out_dir):
    import torch as th
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (one GPU here: RCCL wants one device per rank; gloo moves the same tensors)
    try:
        pl = tspws.Plan(tspws.resolve(abi.default_params(**kw), N), N)
        first, count = tspws.shard_range(mtr, rank, world)
        X = tspws.synth(count, N, seed=31, first=first)
        ls, ts = tspws.stack_sharded(pl, X, first, mtr, schedule="sharded-finish")   # (TSPWS_SHARD_FINISH=0: falls back to "split")
        th.cuda.synchronize()
        np.save(os.path.join(out_dir, f"ls{rank}.npy"), ls.cpu().numpy())
        np.save(os.path.join(out_dir, f"ts{rank}.npy"), ts.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("shard_finish,world", [("1", 2), ("0", 2), ("1", 5)])
def test_two_processes_share_the_gpu_over_gloo(lib, torch, tmp_path, monkeypatch, shard_finish, world):
    """The multi-GPU orchestration (stack_sharded: two-piece streaming + reductions, scale-sharded or redundant finish) with
    the real engine: two processes on this GPU, gloo as the collective.  Every rank must end with the one-process outputs."""
    import socket
    import torch.multiprocessing as mp
    kw, mtr, N = dict(Kmax=10, unbiased=1), 203, 16384   # (203 traces: ragged shards; five ranks: shares of 2-3 octaves)
    monkeypatch.setenv("TSPWS_SHARD_FINISH", shard_finish)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_shard_worker, args=(world, port, kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    pl = tspws.Plan(tspws.resolve(abi.default_params(**kw), N), N)
    ls, ts = pl.stack(tspws.synth(mtr, N, seed=31))
    torch.cuda.synchronize()
    for r in range(world):
        assert abi.relerr(np.load(tmp_path / f"ls{r}.npy"), ls.cpu().numpy()) < 1e-6
        assert abi.relerr(np.load(tmp_path / f"ts{r}.npy"), ts.cpu().numpy()) < 1e-6


@pytest.mark.gpu
def test_frame_without_scales_fails_like_the_reference(lib):
    """fmin above the first scale resolves J to 0: the reference's coefficient containers cannot be created and tspws_main
    returns 4 (FWTa/wavelet_mem_v7.c:40-45, ts_pws1f_lib.c:199-204) -- after fold / mean removal have rewritten the
    traces.  Found by tools/random_sweep.py (the oracle used to return 0 here)."""
    kw = dict(type=-3, s0=4.823433067845736, fmin=0.04796741189352445, wu=1.5, Kmax=12, lrm=1)
    X = abi.synth_traces(9, 1000, seed=3) + np.float32(0.25)
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X)
    assert a["params"].J == 0 and b["params"].J == 0
    assert a["rc"] == 4 and b["rc"] == 4
    np.testing.assert_array_equal(a["sigall"], b["sigall"])
    assert not np.array_equal(a["sigall"], X)           # the mean WAS removed
    assert not a["ls"].any() and not a["tsPWS"].any()   # outputs untouched
    ref = abi.ref()
    if ref is not None:
        r = abi.run_main(ref.tspws_main, p, X)
        assert r["rc"] == 4
        np.testing.assert_array_equal(a["sigall"], r["sigall"])


def _gpu_jk_worker(rank, world, port, kw, mtr, N, out_dir):
    import torch as th
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = tspws.resolve(abi.default_params(**kw), N)
        pl = tspws.Plan(p, N)
        Cn = abi.binomial(p.jackknife_n, p.jackknife_d)
        times = (1262304000 + 86400 * np.sort(np.random.default_rng(4).integers(0, 2 * 365, mtr))).astype(np.int64)
        sel = np.zeros((Cn, mtr), np.int8)
        assert tspws.load().tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, p.jackknife_d, p.jackknife_n, Cn) == 0
        first, count = tspws.shard_range(mtr, rank, world)
        X = tspws.synth(count, N, seed=41, first=first)
        ls, ts, jl, jt, jm = tspws.jackknife_sharded(pl, X, sel, first, mtr)
        th.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"jk{rank}.npz"), ls=ls.cpu().numpy(), ts=ts.cpu().numpy(), jl=jl.cpu().numpy(), jt=jt.cpu().numpy(), jm=jm)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_jackknife_three_processes_over_gloo(lib, torch, tmp_path):
    """jackknife_sharded with the real engine: three processes on this GPU (gloo as the collective), six replicas -> two per
    rank; every rank must end with the oracle's main stack and replicas."""
    import socket
    import torch.multiprocessing as mp
    kw, mtr, N, world = dict(Kmax=4, unbiased=1, jackknife_n=4, jackknife_d=2), 151, 4096, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_jk_worker, args=(world, port, kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    times = (1262304000 + 86400 * np.sort(np.random.default_rng(4).integers(0, 2 * 365, mtr))).astype(np.int64)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), tspws.synth(mtr, N, seed=41).cpu().numpy(), times=times)
    for r in range(world):
        got = np.load(tmp_path / f"jk{r}.npz")
        np.testing.assert_array_equal(got["jm"], want["jk_mtr"])
        assert abi.relerr(got["ls"], want["ls"]) < TOL32 and abi.relerr(got["ts"], want["tsPWS"]) < TOL32
        for c in range(len(want["jk_mtr"])):
            assert abi.relerr(got["jl"][c], want["jk_ls"][c]) < TOL32 and abi.relerr(got["jt"][c], want["jk_ts"][c]) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("N,mtr,Cn,K,pflip", [
    (2052, 157, 3, 5, 0.5),      # 4 | N, not 1024 | N: the last column block has ONE live thread; runs of 1-3 traces: several run ends per batch of 8 rows
    (1028, 90, 14, 4, 0.3),      # 14 replicas: the 16-column instantiation of the walk (128 KB of LDS for the writer wave)
    (4096, 300, 10, 10, 0.02),   # long runs with a few short ones in between
    (3000, 131, 5, 7, 0.5),      # N % 4 == 0, ragged; Kmax does not divide anything
    (1501, 77, 4, 3, 0.4),       # odd N: the scalar form of the walk (no writer wave)
])
def test_one_pass_rows_of_arbitrary_selections(lib, torch, N, mtr, Cn, K, pflip):
    """The rows of the one-pass walk (k_rows_walk: running sums per column, flushes through LDS to the writer wave, one load stream over the
    run ends) for ARBITRARY selections -- not only the jackknife's day bins -- against sums formed here: row (c, g) = the traces k of replica c
    with floor(k_sel K / K_c) = g (ts_pws1f_lib.c:766), the plain groups min(floor(i K / mtr), K - 1) (:876)."""
    p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1, jackknife_n=6, jackknife_d=1), N)
    pl = tspws.Plan(p, N)
    X = abi.synth_traces(mtr, N, seed=31 + Cn)
    rng = np.random.default_rng(100 + N)
    sel = np.ones((Cn, mtr), np.int8)
    for c in range(Cn):  # a 0/1 sequence that flips with probability pflip per trace
        flips = rng.random(mtr) < pflip
        sel[c] = (np.cumsum(flips) + c) % 2
    sel[0, :] = 1 if Cn > 1 else sel[0]  # one replica that keeps every trace
    Xd = torch.as_tensor(X, device="cuda")
    pl.jackknife_local(Xd, 0, mtr, sel)
    torch.cuda.synchronize()
    rows = pl.jackknife_buffer(Cn).cpu().numpy().reshape(Cn, K, N)
    main = pl.reduce_buffer(mtr).cpu().numpy().reshape(K, N)
    X64 = X.astype(np.float64)
    want_main = np.zeros((K, N))
    for i in range(mtr):
        want_main[min(i * K // mtr, K - 1)] += X64[i]
    scale = np.abs(want_main).max()
    assert np.abs(main - want_main).max() <= 1e-12 * scale
    for c in range(Cn):
        idx = np.flatnonzero(sel[c] == 1)
        want = np.zeros((K, N))
        for k, i in enumerate(idx):
            want[min(k * K // max(len(idx), 1), K - 1)] += X64[i]
        assert np.abs(rows[c] - want).max() <= 1e-12 * scale, c
