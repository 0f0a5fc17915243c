"""The host C of the drop-in (command line ts_pws_cli.c, SAC / msacs reader and writer sacio_min.c, tspws_main.c) under AddressSanitizer +
UBSan + LeakSanitizer (`make -C ts-pws_amd asan`), over valid and malformed inputs.  No GPU needed: the readers run before the first HIP
call, which then ends the run with status 5 ("no usable HIP device") on a CPU box -- on a GPU box the same runs go through and write
their outputs into the temporary directory.  Reference reader: /root/reference/src/ts_pws1f.c:570-723 (msacs header :586-608)."""
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "ts-pws_amd")
BIN = os.path.join(PKG, "bin", "ts_pws_asan")
SAC = sorted(os.path.join(HERE, "golden", "sac", f) for f in os.listdir(os.path.join(HERE, "golden", "sac")) if f.endswith(".sac"))


@pytest.fixture(scope="module")
def asan_bin():
    if not os.path.exists(os.path.join(PKG, "lib", "libtspws_hip.so")):
        pytest.skip("library not built (run __graft_entry__.build())")
    r = subprocess.run(["make", "-C", PKG, "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(BIN), r.stdout + r.stderr
    return BIN


def run(asan_bin, cwd, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([asan_bin, *args], cwd=cwd, capture_output=True, text=True, env=env, timeout=600)
    bad = [l for l in (r.stdout + r.stderr).splitlines() if "AddressSanitizer" in l or "LeakSanitizer" in l or "runtime error" in l]
    assert not bad and r.returncode != 99, (args, r.returncode, bad, r.stderr[-2000:])
    return r.returncode if r.returncode < 128 else r.returncode - 256, r.stdout


def stacked(rc):
    """status of a run whose input was accepted: 0 where a GPU did the stacking, 5 (no usable HIP device) on a CPU box"""
    return rc in (0, 5)


def msacs(path, traces, times=None, cut=None, nlags=None, nseq=None):
    """the reference's container (src/sac2bin.h:6-27): 116-byte header, time_t[nseq], float lag0[nseq], float data[nseq][nlags]"""
    mtr, n = traces.shape
    hdr = b"".join(s.ljust(8, b"\0") for s in (b"cc", b"N1", b"STA1", b"00", b"BHZ", b"N2", b"STA2", b"00", b"BHZ"))
    hdr += struct.pack("<6f", 1, 2, 3, 4, 5, 6) + struct.pack("<2I", n if nlags is None else nlags, mtr if nseq is None else nseq)
    hdr += struct.pack("<3f", float(n - 1), -0.5 * (n - 1), 0.5 * (n - 1))
    assert len(hdr) == 116
    t = (np.arange(mtr, dtype=np.int64) * 86400 + 1262304000) if times is None else times
    blob = hdr + t.astype("<i8").tobytes() + np.zeros(mtr, "<f4").tobytes() + traces.astype("<f4").tobytes()
    with open(path, "wb") as f:
        f.write(blob if cut is None else blob[:cut])


def test_valid_sac_list_and_options(asan_bin, tmp_path):
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(SAC) + "\n")
    for opts in ([], ["rm", "fold", "fmin=0.004", "J=3", "verbose"], ["TwoStage=2", "unbiased", "osac=x"], ["MexHat", "convergence", "AllSteps"],
                 ["wu=notanumber", "J=-3", "Nmax=2"]):
        rc, out = run(asan_bin, tmp_path, str(lst), *opts)
        assert stacked(rc), (opts, rc, out)
    assert run(asan_bin, tmp_path)[0] == 0 and run(asan_bin, tmp_path, "info")[0] == 0


def test_malformed_sac_inputs(asan_bin, tmp_path):
    good = open(SAC[0], "rb").read()
    (tmp_path / "hdr_cut.sac").write_bytes(good[:400])                       # inside the 632-byte header
    (tmp_path / "data_cut.sac").write_bytes(good[:3000])                     # inside the samples
    zero = bytearray(good); zero[70 * 4 + 9 * 4: 70 * 4 + 10 * 4] = struct.pack("<i", 0)
    (tmp_path / "npts0.sac").write_bytes(bytes(zero))                        # npts = 0
    neg = bytearray(good); neg[70 * 4 + 9 * 4: 70 * 4 + 10 * 4] = struct.pack("<i", -7)
    (tmp_path / "nptsneg.sac").write_bytes(bytes(neg))
    (tmp_path / "garbage.sac").write_bytes(np.random.default_rng(1).bytes(5000))
    cases = {
        "missing_list": (["no_such_list.txt"], -2),
        "empty_list": (["empty.txt"], 0),
        "missing_file": (["l_missing.txt"], 2),
        "hdr_cut": (["l_hdr_cut.txt"], 2),
        "npts0": (["l_npts0.txt"], 2),
        "nptsneg": (["l_nptsneg.txt"], 2),
        "garbage": (["l_garbage.txt"], 2),
        "second_cut": (["l_second_cut.txt"], -2),                            # the first file sizes the ensemble, the second one is short
        "second_missing": (["l_second_missing.txt"], -2),
    }
    (tmp_path / "empty.txt").write_text("")
    (tmp_path / "l_missing.txt").write_text("/no/such/file.sac\n")
    for k in ("hdr_cut", "npts0", "nptsneg", "garbage"):
        (tmp_path / f"l_{k}.txt").write_text(str(tmp_path / f"{k}.sac") + "\n")
    (tmp_path / "l_second_cut.txt").write_text(SAC[0] + "\n" + str(tmp_path / "data_cut.sac") + "\n")
    (tmp_path / "l_second_missing.txt").write_text(SAC[0] + "\n/no/such/file.sac\n")
    for name, (args, want) in cases.items():
        rc, out = run(asan_bin, tmp_path, *args)
        assert rc == want, (name, rc, out)


def test_msacs_container_inputs(asan_bin, tmp_path):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((6, 301)).astype(np.float32)
    msacs(tmp_path / "ok.bin", X)
    rc, out = run(asan_bin, tmp_path, "ok.bin", "bin", "TwoStage=3", "jackknife_n=3", "jackknife_d=1")
    assert stacked(rc), (rc, out)
    # cut inside the data block: the reference's warning, the rest of the rows stay zero, the run goes on (ts_pws1f.c:664-675)
    msacs(tmp_path / "data_cut.bin", X, cut=116 + 6 * 12 + 3 * 301 * 4 + 10)
    rc, out = run(asan_bin, tmp_path, "data_cut.bin", "bin")
    assert stacked(rc) and "shorter than predicted (data)" in out, (rc, out)
    # short header: the reference prints this and reads on into an uninitialised header; here the ensemble ends
    (tmp_path / "short.bin").write_bytes(open(tmp_path / "ok.bin", "rb").read()[:60])
    rc, out = run(asan_bin, tmp_path, "short.bin", "bin")
    assert rc == -2 and "shorter than predicted (header)" in out
    # headers that promise what the file cannot hold, a single lag (dt = x / 0), random bytes
    msacs(tmp_path / "huge.bin", X, nseq=4000000000)
    msacs(tmp_path / "wide.bin", X, nlags=2000000000)
    msacs(tmp_path / "onelag.bin", X, nlags=1)
    msacs(tmp_path / "tables_cut.bin", X, cut=116 + 20)
    (tmp_path / "garbage.bin").write_bytes(rng.bytes(5000))
    for name in ("huge", "wide", "onelag", "tables_cut", "garbage"):
        rc, out = run(asan_bin, tmp_path, name + ".bin", "bin")
        assert rc == 2 and "header corrupted" in out, (name, rc, out)
    msacs(tmp_path / "none.bin", X[:0])                                     # nseq = 0: nothing to do
    assert run(asan_bin, tmp_path, "none.bin", "bin")[0] == 0
    assert run(asan_bin, tmp_path, "missing.bin", "bin")[0] == -2


def test_batch_of_ensembles(asan_bin, tmp_path):
    """`ts_pws @batch`: every ensemble is tried, the first non-zero status is the exit code, nothing leaks when one of them is malformed."""
    (tmp_path / "a.txt").write_text("\n".join(SAC) + "\n")
    (tmp_path / "b.txt").write_text("\n".join(SAC[:2]) + "\n")
    (tmp_path / "bad.txt").write_text(SAC[0] + "\n/no/such/file.sac\n")
    (tmp_path / "batch.txt").write_text("# station pairs\na.txt\n\nbad.txt pair_bad\nb.txt pair_b\n")
    rc, out = run(asan_bin, tmp_path, "@batch.txt", "TwoStage=2")
    assert "ensemble 1 (bad.txt) ended with status -2" in out
    assert rc in (-2, 5), (rc, out)      # (CPU box: the first failure is ensemble 0's missing device)
    assert run(asan_bin, tmp_path, "@no_such_batch.txt")[0] == -2
    (tmp_path / "empty_batch.txt").write_text("\n# nothing\n")
    assert run(asan_bin, tmp_path, "@empty_batch.txt")[0] == 0
