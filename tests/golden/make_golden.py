#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref/libtspws_ref.so, which
oracle/Makefile compiles in place from /root/reference/src (no reference source
is copied).  The reference ships no expected outputs, so
these arrays -- inputs from this repo's own seeded generator, outputs from the
reference library -- are what pins the oracle and the HIP path.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Fixtures are data only: inputs and expected outputs.
"""
import ctypes as C
import glob
import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"  # the reference draws random-subsampling masks inside an OpenMP loop: keep the order fixed

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import abi  # noqa: E402

REF_EXAMPLES = "/root/reference/examples/ECH.00Z.CAN.00Z_500days"


class t_WaveletFamily(C.Structure):  # layout of FWTa/wavelet_v7.h:36-56
    _fields_ = [
        ("format", C.c_int), ("type", C.c_int), ("convtype", C.c_int),
        ("wframe", C.POINTER(C.c_void_p)), ("wdualframe", C.POINTER(C.c_void_p)),
        ("scale", C.POINTER(C.c_double)), ("Ls", C.POINTER(C.c_uint)), ("Lds", C.POINTER(C.c_uint)),
        ("center", C.POINTER(C.c_int)), ("center_df", C.POINTER(C.c_int)), ("Down_smp", C.POINTER(C.c_uint)),
        ("Ns", C.c_uint), ("V", C.c_uint), ("Cpsi", C.c_double), ("a0", C.c_double), ("b0", C.c_double), ("op1", C.c_double),
    ]


class t_CWTvar(C.Structure):  # FWTa/wavelet_v7.h:64-68
    _fields_ = [("d", C.POINTER(C.c_void_p)), ("N", C.POINTER(C.c_uint)), ("S", C.c_uint)]


def ref_lib():
    lib = abi.ref()
    if lib is None:
        sys.exit("oracle/_ref/libtspws_ref.so missing: run `make -C oracle ref` in the build container")
    lib.CreateWaveletFamily.restype = C.POINTER(t_WaveletFamily)
    lib.CreateWaveletFamily.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]
    lib.DestroyWaveletFamily.argtypes = [C.POINTER(t_WaveletFamily)]
    lib.CreateComplexWaveletVar.restype = C.POINTER(t_CWTvar)
    lib.CreateComplexWaveletVar.argtypes = [C.POINTER(t_WaveletFamily), C.c_uint]
    lib.DestroyComplexWaveletVar.argtypes = [C.POINTER(t_CWTvar)]
    lib.complex_1D_wavelet_dec.argtypes = [C.POINTER(t_CWTvar), C.c_void_p, C.c_uint, C.POINTER(t_WaveletFamily)]
    lib.Re_complex_1D_wavelet_rec.argtypes = [C.c_void_p, C.POINTER(t_CWTvar), C.c_uint, C.POINTER(t_WaveletFamily)]
    return lib


def frame_case(lib, name, params, N, edge=4):
    """Resolve parameters with the REFERENCE (a 0-trace tspws_main call mutates the
    struct exactly like a real one), then dump the family tables."""
    p = abi.t_tsPWS.from_buffer_copy(params)
    out = abi.t_tsPWS_out()
    d = abi.t_data()
    d.hdr.max, d.hdr.mtr, d.hdr.dt, d.hdr.beg = N, 0, 1.0, 0.0
    assert lib.tspws_main(C.byref(p), C.byref(out), C.byref(d)) == 0
    wf = lib.CreateWaveletFamily(p.type, p.J, p.V, N, p.s0, p.b0, 0, p.w0, p.uni)
    f = wf.contents
    S = f.Ns
    tab = dict(
        J=p.J, V=p.V, s0=p.s0, b0=p.b0, w0=p.w0, N=N, type=p.type, S=S, Cpsi=f.Cpsi,
        scale=np.array([f.scale[s] for s in range(S)]),
        L=np.array([f.Ls[s] for s in range(S)], np.uint32),
        c=np.array([f.center[s] for s in range(S)], np.int32),
        cd=np.array([f.center_df[s] for s in range(S)], np.int32),
        D=np.array([f.Down_smp[s] for s in range(S)], np.uint32),
    )
    head = np.zeros((S, edge), np.complex128)
    tail = np.zeros((S, edge), np.complex128)
    dhead = np.zeros((S, edge), np.complex128)
    csum = np.zeros(S, np.complex128)
    for s in range(S):
        L = int(f.Ls[s])
        w = np.ctypeslib.as_array(C.cast(f.wframe[s], C.POINTER(C.c_double)), shape=(2 * L,)).view(np.complex128)
        wd = np.ctypeslib.as_array(C.cast(f.wdualframe[s], C.POINTER(C.c_double)), shape=(2 * L,)).view(np.complex128)
        head[s], tail[s], dhead[s], csum[s] = w[:edge], w[-edge:], wd[:edge], w.sum()
    tab.update(w_head=head, w_tail=tail, wd_head=dhead, w_sum=csum)
    lib.DestroyWaveletFamily(wf)
    return {f"{name}/{k}": v for k, v in tab.items()}


def cwt_case(lib, name, params, x):
    N = len(x)
    p = abi.t_tsPWS.from_buffer_copy(params)
    out = abi.t_tsPWS_out()
    d = abi.t_data()
    d.hdr.max, d.hdr.mtr, d.hdr.dt = N, 0, 1.0
    lib.tspws_main(C.byref(p), C.byref(out), C.byref(d))
    wf = lib.CreateWaveletFamily(p.type, p.J, p.V, N, p.s0, p.b0, 0, p.w0, p.uni)
    wt = lib.CreateComplexWaveletVar(wf, N)
    xd = np.ascontiguousarray(x, np.float64)
    assert lib.complex_1D_wavelet_dec(wt, xd.ctypes.data, N, wf) == 0
    S = wt.contents.S
    Ns = [int(wt.contents.N[s]) for s in range(S)]
    Y = np.concatenate([
        np.ctypeslib.as_array(C.cast(wt.contents.d[s], C.POINTER(C.c_double)), shape=(2 * Ns[s],)).view(np.complex128).copy()
        for s in range(S)])
    xr = np.zeros(N)
    assert lib.Re_complex_1D_wavelet_rec(xr.ctypes.data, wt, N, wf) == 0
    lib.DestroyComplexWaveletVar(wt)
    lib.DestroyWaveletFamily(wf)
    return {f"{name}/x": xd, f"{name}/Y": Y, f"{name}/xrec": xr, f"{name}/J": p.J, f"{name}/V": p.V,
            f"{name}/s0": p.s0, f"{name}/b0": p.b0, f"{name}/w0": p.w0, f"{name}/type": p.type}


PARAM_KEYS = ["type", "uni", "J", "V", "s0", "b0", "w0", "wu", "fmin", "Q", "cycle", "w0set", "lrm", "lVfix", "ls0fix",
              "lb0fix", "fold", "unbiased", "jackknife_n", "jackknife_d", "Nmax", "Kmax", "convergence", "AllSteps",
              "subsmpl_N", "subsmpl_p"]


def params_to_arrays(prefix, p):
    return {f"{prefix}/{k}": getattr(p, k) for k in PARAM_KEYS}


def main_case(lib, name, params, traces, dt=1.0, beg=0.0, times=None, reference=None):
    abi.srand(1)
    r = abi.run_main(lib.tspws_main, params, traces, dt=dt, beg=beg, times=times, reference=reference)
    assert r["rc"] == 0, (name, r["rc"])
    d = {f"{name}/ls": r["ls"], f"{name}/tsPWS": r["tsPWS"], f"{name}/dt": dt, f"{name}/beg": beg}
    d.update(params_to_arrays(f"{name}/in", params))
    d.update(params_to_arrays(f"{name}/out", r["params"]))
    if params.lrm or params.fold:
        d[f"{name}/sigall_after"] = r["sigall"]
    for k in [k for k in r if k.startswith("conv_") or k.startswith("sub_")]:
        d[f"{name}/{k}"] = r[k]
    if reference is not None:
        d[f"{name}/reference"] = np.asarray(reference, np.float32)
    if "jk_ls" in r:
        d.update({f"{name}/jk_ls": r["jk_ls"], f"{name}/jk_ts": r["jk_ts"], f"{name}/jk_mtr": r["jk_mtr"],
                  f"{name}/times": np.asarray(times, np.int64)})
    return d


def read_sac_data(path):
    """Little-endian SAC v6: 632-byte header (70 f32, 40 i32, 192 chars) then npts f32."""
    raw = open(path, "rb").read()
    fl = np.frombuffer(raw, "<f4", 70, 0)
    it = np.frombuffer(raw, "<i4", 40, 280)
    npts = int(it[9])
    return float(fl[0]), float(fl[5]), np.frombuffer(raw, "<f4", npts, 632).copy()


def main():
    lib = ref_lib()
    P = abi.default_params
    # (1) frame tables -------------------------------------------------------
    frames = {}
    for N in (2048, 16501, 131072):
        frames.update(frame_case(lib, f"morlet_N{N}", P(), N))
    frames.update(frame_case(lib, "morlet_w2pi_N32768", P(w0=2 * np.pi), 32768))
    for N in (2048, 131072):
        frames.update(frame_case(lib, f"mexhat_N{N}", P(type=-3), N))
    frames.update(frame_case(lib, "exact_morlet_N2048", P(type=-2), 2048))
    frames.update(frame_case(lib, "morlet_fmin_J3_N16501", P(fmin=0.004 * 4.0, J=3), 16501))  # dt=1 here
    frames.update(frame_case(lib, "morlet_Q4_N4096", P(Q=4.0, w0set=1), 4096))
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)

    # (2) single-trace forward / inverse ---------------------------------------
    cwt = {}
    cwt.update(cwt_case(lib, "morlet_N2048", P(), abi.synth_traces(1, 2048, seed=11)[0]))
    cwt.update(cwt_case(lib, "morlet_N1501", P(), abi.synth_traces(1, 1501, seed=12)[0]))
    cwt.update(cwt_case(lib, "mexhat_N2048", P(type=-3), abi.synth_traces(1, 2048, seed=13)[0]))
    cwt.update(cwt_case(lib, "morlet_oddD_N1501", P(s0=3.7, J=4), abi.synth_traces(1, 1501, seed=14)[0]))
    np.savez_compressed(os.path.join(HERE, "cwt.npz"), **cwt)

    # (3) whole calls on 16 x 2048 ---------------------------------------------
    X = abi.synth_traces(16, 2048, seed=3)
    Xodd = abi.synth_traces(16, 2047, seed=4)
    Xz = X.copy()
    Xz[5] = 0.0
    times = 1262304000 + 86400 * 23 * np.arange(16)  # 2010-01-01 + 23-day steps: spreads over the year
    cases = {
        "single_wu2": (P(), X, {}),
        "single_wu1": (P(wu=1.0), X, {}),
        "single_wu1p5": (P(wu=1.5), X, {}),
        "single_unbiased": (P(unbiased=1), X, {}),
        "two_K4_biased": (P(Kmax=4), X, {}),
        "two_K4_unbiased": (P(Kmax=4, unbiased=1), X, {}),
        "two_K4_wu1p5": (P(Kmax=4, wu=1.5), X, {}),
        "two_K1_unbiased": (P(Kmax=1, unbiased=1), X, {}),
        "two_K16_unbiased": (P(Kmax=16, unbiased=1), X, {}),
        "Kmax_gt_mtr": (P(Kmax=17, unbiased=1), X, {}),
        "rm": (P(lrm=1), X + np.float32(0.3), {}),
        "fold_even": (P(fold=1), X, dict(beg=-1023.5)),
        "fold_odd": (P(fold=1), Xodd, dict(beg=-1023.0)),
        "fold_ignored": (P(fold=1), X, dict(beg=0.0)),
        "rm_fold_two": (P(lrm=1, fold=1, Kmax=4, unbiased=1), Xodd + np.float32(0.1), dict(beg=-1023.0)),
        "mexhat": (P(type=-3), X, {}),
        "mexhat_two": (P(type=-3, Kmax=4, unbiased=1), X, {}),
        "exact_morlet": (P(type=-2), X, {}),
        "w2pi": (P(w0=2 * np.pi), X, {}),
        "zero_trace": (P(), Xz, {}),
        "zero_trace_two": (P(Kmax=16), Xz, {}),
        "Nmax8": (P(Nmax=8), X, {}),
        "uni_J3": (P(uni=1, J=3), X, {}),
        "fmin_J3": (P(fmin=0.02, J=3), X, {}),
        "fmin_only": (P(fmin=0.01), X, {}),
        "oddD": (P(s0=3.7, J=4), Xodd, {}),
        "cycles3": (P(cycle=3.0, w0set=2), X, {}),
        "jk_n4_d1": (P(Kmax=4, jackknife_n=4, jackknife_d=1), X, dict(times=times)),
        "jk_n5_d2": (P(Kmax=4, unbiased=1, jackknife_n=5, jackknife_d=2), X, dict(times=times)),
        "jk_mexhat": (P(type=-3, Kmax=2, jackknife_n=3, jackknife_d=1), X, dict(times=times)),
        "conv_single_steps": (P(convergence=1, AllSteps=1), X, {}),
        "conv_two_K4": (P(convergence=1, Kmax=4, unbiased=1), X, {}),
        "conv_reference": (P(convergence=1, wu=1.0), X, dict(reference=X[3])),
        "sub_single": (P(subsmpl_N=3, subsmpl_p=0.5), X, {}),
        "sub_single_dense": (P(subsmpl_N=2, subsmpl_p=0.8, unbiased=1), X, {}),
        "sub_two_K4": (P(subsmpl_N=3, subsmpl_p=0.6, Kmax=4, unbiased=1), X, {}),
    }
    mains = {"X": X, "Xodd": Xodd}
    inputs = {}
    for name, (p, x, kw) in cases.items():
        mains.update(main_case(lib, name, p, x, **kw))
        inputs[name] = x
        if x is X:
            mains[f"{name}/input"] = "X"
        elif x is Xodd:
            mains[f"{name}/input"] = "Xodd"
        else:
            mains[f"{name}/input"] = "own"
            mains[f"{name}/x"] = x
    np.savez_compressed(os.path.join(HERE, "mains.npz"), **mains)

    # (4) shipped example data: first 32 daily correlations -----------------------
    files = sorted(glob.glob(os.path.join(REF_EXAMPLES, "*.sac")))
    assert len(files) == 499
    ex = {}
    tr = []
    for f in files[:32]:
        dt, b, x = read_sac_data(f)
        tr.append(x)
    E = np.stack(tr)
    ex["traces"] = E
    ex["dt"], ex["beg"] = dt, b
    ex.update(main_case(lib, "ex1", P(), E, dt=dt, beg=b))
    ex.update(main_case(lib, "ex2", P(lrm=1, fold=1, fmin=0.004, J=3), E, dt=dt, beg=b))
    ex.update(main_case(lib, "ex3", P(lrm=1, fold=1, fmin=0.004, J=3, Kmax=10, unbiased=1), E, dt=dt, beg=b))
    ex.update(main_case(lib, "ex_mexhat", P(type=-3), E, dt=dt, beg=b))
    for k in [k for k in ex if k.endswith("sigall_after")]:
        del ex[k]
    np.savez_compressed(os.path.join(HERE, "example32.npz"), **ex)
    print("golden fixtures written to", HERE)


def extra():
    """Round 4 additions, in a file of their own (the fixtures above stay byte for byte what they were)."""
    lib = ref_lib()
    P = abi.default_params
    X = abi.synth_traces(16, 2048, seed=3)
    # leap-day quirk of JackknifePlans (ts_pws1f_lib.c:398-401): tm_yday == 365 (31 December of a leap year) gives bin == n,
    # which no combination deletes -- traces 14 and 15 (31 Dec 2012, 31 Dec 2016) stay in EVERY replica
    times = 1325376000 + 86400 * 23 * np.arange(16)          # 2012-01-01 + 23-day steps
    times[14] = 1325376000 + 86400 * 365                      # 2012-12-31: day 366 of a leap year
    times[15] = 1451606400 + 86400 * 365                      # 2016-12-31
    ex = {"X": X}
    for name, p in (("jk_leapday_n4_d1", P(Kmax=4, jackknife_n=4, jackknife_d=1)), ("jk_leapday_n6_d2", P(Kmax=3, unbiased=1, jackknife_n=6, jackknife_d=2))):
        ex.update(main_case(lib, name, p, X, times=times))
        ex[f"{name}/input"] = "X"
    np.savez_compressed(os.path.join(HERE, "extra.npz"), **ex)
    print("extra fixtures written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
    else:
        main()
