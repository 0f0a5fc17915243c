"""End-to-end test of the SAC command-line front-end on the GPU: the three invocations of the reference's
examples/example.sh (on the first 32 shipped traces) against the golden outputs of the reference library."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

import abi

pytestmark = pytest.mark.gpu
tspws = importlib.import_module("ts-pws_amd")
EXE = os.path.join(abi.ROOT, "ts-pws_amd", "bin", "ts_pws")


@pytest.fixture(scope="module")
def sac_list(tmp_path_factory, golden):
    d = tmp_path_factory.mktemp("sac")
    g = golden["example32"]
    names = []
    for i, x in enumerate(g["traces"]):
        p = d / f"t{i:03d}.sac"
        abi.write_sac(str(p), x, float(g["dt"]), float(g["beg"]), year=2010, jday=1 + 11 * i, kstnm="CAN")
        names.append(str(p))
    (d / "list.txt").write_text("\n".join(names) + "\n")
    return d


def run_cli(cwd, *args):
    r = subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_example1_default(sac_list, golden):
    g = golden["example32"]
    out = run_cli(sac_list, "list.txt", "verbose")
    assert "V = 4, J = 11" in out
    ls, ts = abi.read_sac(sac_list / "tl.sac"), abi.read_sac(sac_list / "ts_pws.sac")
    assert abi.relerr(ls["data"], g["ex1/ls"]) < 2e-6 and abi.relerr(ts["data"], g["ex1/tsPWS"]) < 2e-6
    assert ts["i"][9] == 16501 and ts["f"][0] == 4.0 and ts["f"][5] == -33000.0 and ts["f"][40] == 32.0  # user0 = mtr
    assert ts["k"][184:192] == b"ts_pws  " and ls["k"][184:192] == b"t-lin   " and ts["i"][17] == 11 and ts["i"][38] == 1


def test_example2_fold_trim(sac_list, golden):
    g = golden["example32"]
    run_cli(sac_list, "list.txt", "osac=example2", "wu=2", "rm", "fold", "fmin=0.004", "J=3")
    ts = abi.read_sac(sac_list / "ts_pws_example2.sac")
    ls = abi.read_sac(sac_list / "tl_example2.sac")
    # folded outputs keep samples max/2 .. max-1 and start at b + dt*(max/2)   (ts_pws1f.c:322-328)
    assert ts["i"][9] == 8251 and ts["f"][5] == 0.0
    assert abi.relerr(ts["data"], g["ex2/tsPWS"][8250:]) < 2e-6 and abi.relerr(ls["data"], g["ex2/ls"][8250:]) < 2e-6


def test_example3_bin_twostage_and_jackknife(sac_list, golden):
    g = golden["example32"]
    X = g["traces"]
    mtr, n = X.shape
    # msacs container (sac2bin.c:192-195): 116-byte header, time_t[mtr], float lag0[mtr], float data[mtr][n]
    hdr = struct.pack("<8s8s8s8s8s8s8s8s8s6fII3f", b"ccgn", b"", b"ECH", b"00", b"Z", b"", b"CAN", b"00", b"Z",
                      48.2, 7.15, 0.0, -35.3, 149.0, 0.0, n, mtr, float(g["dt"]) * (n - 1), float(g["beg"]),
                      float(g["beg"]) + float(g["dt"]) * (n - 1))
    assert len(hdr) == 116
    times = (1262304000 + 86400 * 11 * np.arange(mtr)).astype(np.int64)
    with open(sac_list / "ens.bin", "wb") as f:
        f.write(hdr + times.tobytes() + np.zeros(mtr, np.float32).tobytes() + X.astype(np.float32).tobytes())
    run_cli(sac_list, "ens.bin", "osac=twostage", "wu=2", "rm", "bin", "fold", "TwoStage=10", "unbiased", "fmin=0.004", "J=3")
    ts = abi.read_sac(sac_list / "ts_pws_twostage.sac")
    assert abi.relerr(ts["data"], g["ex3/tsPWS"][8250:]) < 2e-6
    # jackknife through the CLI: replicas must match the oracle fed with the same start times
    run_cli(sac_list, "ens.bin", "bin", "osac=jk", "TwoStage=4", "jackknife_n=4", "jackknife_d=1", "obin")
    p = abi.default_params(Kmax=4, jackknife_n=4, jackknife_d=1)
    want = abi.run_main(abi.oracle().orc_tspws_main, p, X, dt=float(g["dt"]), beg=float(g["beg"]), times=times)
    raw = open(sac_list / "ts_pws_jk_subsmpl.bin", "rb").read()
    M = 4
    sizes = np.frombuffer(raw, "<i8", M, 116)
    rows = np.frombuffer(raw, "<f4", M * n, 116 + 8 * M).reshape(M, n)   # wrbin writes no lag0 block (ts_pws1f.c:817-819)
    np.testing.assert_array_equal(sizes, want["jk_mtr"])
    for c in range(M):
        assert abi.relerr(rows[c], want["jk_ts"][c]) < 2e-6
    # the same jackknife from the SAC list (start times from the SAC reference time -- this front-end's extension)
    run_cli(sac_list, "list.txt", "osac=jk2", "TwoStage=4", "jackknife_n=4", "jackknife_d=1")
    r0 = abi.read_sac(sac_list / "ts_pws_jk2_subsmpl_0.sac")
    assert abi.relerr(r0["data"], want["jk_ts"][0]) < 2e-6 and r0["f"][40] == float(want["jk_mtr"][0])


def test_convergence_and_subsampling_outputs(sac_list, golden):
    g = golden["example32"]
    run_cli(sac_list, "list.txt", "osac=cv", "convergence", "AllSteps", "subsmpl_N=2", "subsmpl_prob=0.5")
    sim = np.fromfile(sac_list / "ts_pws_cv_convergence", "<f8")
    mis = np.fromfile(sac_list / "ts_pws_cv_misfit", "<f8")
    steps = np.fromfile(sac_list / "ts_pws_cv_steps", "<f4").reshape(32, 16501)
    assert sim.shape == (32,) and abs(sim[-1] - 1.0) < 1e-6 and mis.shape == (32,) and mis[-1] < 1e-6 * mis[0]
    assert abi.relerr(steps[-1], g["ex1/tsPWS"]) < 2e-6
    s0 = abi.read_sac(sac_list / "ts_pws_cv_subsmpl_0.sac")
    assert s0["i"][9] == 16501 and np.isfinite(s0["data"]).all() and s0["data"].any()


def test_nmax_beyond_the_trace_count_is_clamped(sac_list, golden):
    """Nmax larger than the traces read: tspws_main clamps it, and the CLI sizes its convergence dumps and the user0 header
    field with the clamped count (the reference takes Nmax unchecked, ts_pws1f_lib.c:65)."""
    g = golden["example32"]
    out = run_cli(sac_list, "list.txt", "osac=nm", "Nmax=1000", "convergence")
    assert "exceeds" in out
    ts = abi.read_sac(sac_list / "ts_pws_nm.sac")
    assert ts["f"][40] == 32.0                                                       # user0 = traces actually stacked
    assert abi.relerr(ts["data"], g["ex1/tsPWS"]) < 2e-6
    sim = np.fromfile(sac_list / "ts_pws_nm_convergence", "<f8")
    assert sim.shape == (32,) and abs(sim[-1] - 1.0) < 1e-6
    # Nmax below the count still selects a prefix
    run_cli(sac_list, "list.txt", "osac=n8", "Nmax=8")
    assert abi.read_sac(sac_list / "ts_pws_n8.sac")["f"][40] == 8.0
