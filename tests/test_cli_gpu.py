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


def run_cli(cwd, *args, env=None):
    e = None
    if env:
        e = dict(os.environ)
        e.update(env)
    r = subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True, timeout=600, env=e)
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


def test_cli_over_a_device_list_needs_no_python(sac_list, golden):
    """The C front-end alone (no Python, no torch in the process) under TSPWS_DEVICES: (a) one device through RCCL -- the
    library binds librccl.so.1 by itself and the stacks go through ncclAllReduce on a one-device communicator; (b) three
    virtual shards of the one GPU (the library's own reduction kernel).  Same files as the single-device run."""
    g = golden["example32"]
    for tag, env in (("rccl", dict(TSPWS_DEVICES="0", TSPWS_COMM="rccl")), ("loc3", dict(TSPWS_DEVICES="0,0,0"))):
        run_cli(sac_list, "list.txt", f"osac=md_{tag}", "wu=2", "rm", "fold", "TwoStage=10", "unbiased", "fmin=0.004", "J=3", env=env)
        ts = abi.read_sac(sac_list / f"ts_pws_md_{tag}.sac")
        ls = abi.read_sac(sac_list / f"tl_md_{tag}.sac")
        assert abi.relerr(ts["data"], g["ex3/tsPWS"][8250:]) < 2e-6 and ts["f"][40] == 32.0
        assert np.isfinite(ls["data"]).all() and ls["i"][9] == 8251
        run_cli(sac_list, "list.txt", f"osac=sd_{tag}", env=env)            # single-stage: ST || PS are what the devices add
        ts1 = abi.read_sac(sac_list / f"ts_pws_sd_{tag}.sac")
        assert abi.relerr(ts1["data"], g["ex1/tsPWS"]) < 2e-6


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


def _bench(args, env_extra=None, timeout=600):
    import json
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(abi.ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` without a launcher around it: the parent starts two ranks before touching the GPU (here both on the
    one GPU of the box, collectives over gloo), rank 0's JSON line comes back through the parent."""
    r = _bench(["--gpus", "2", "--traces", "64", "--samples", "4096", "--steps", "3", "--warmup", "1", "--strong-total", "300"], {"BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["config"]["world_size"] == 2 and r["config"]["backend"] == "gloo"
    # the one multi-GPU run is an A/B of the three placements of the reduction (same plan, same traces, `steps` timed steps each); the
    # headline fields are the default schedule's; `single` -- north_star's ONE all-reduce -- ends bit-identical to the one-GPU call
    sch = r["schedules"]
    assert set(sch) == {"single", "split", "sharded-finish"} and sum(1 for v in sch.values() if v["is_default"]) == 1
    for name, v in sch.items():
        assert v["ms_per_step"] > 0 and v["value"] > 0 and len(v["per_rank_ms"]) == 2 and v["runs_as"] in ("single", "split", "sharded-finish"), name
    assert sch["single"]["is_default"] and abs(sch["single"]["ms_per_step"] - r["ms_per_step"]) < 1e-9 and "aborted_leg" not in r
    import hashlib
    import torch
    pl = tspws.Plan(tspws.resolve(abi.default_params(Kmax=10, unbiased=1), 4096), 4096)
    ls, ts = pl.stack(tspws.synth(128, 4096, seed=1))
    torch.cuda.synchronize()
    one = hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:16]
    assert sch["single"]["output_sha1"] == one and sch["split"]["output_sha1"] == one
    st = r["strong_scaling"]
    assert st["traces_total"] == 300 and st["traces_on_rank0"] == 150 and st["scaling"] == "strong" and st["value"] > 0 and len(st["per_rank_ms"]) == 2
    # the N > 1 line is complete: the CPU path on rank 0's shard beside the GPU number (cores stated), the shard's partial stacks
    # checked against it, the roofline object (counter traffic null: collected on one GPU only), where every rank's step went
    for k in ("cpu_baseline", "roofline", "per_rank_ms"):
        assert k in r, k
    cb = r["cpu_baseline"]
    assert cb["rc"] == 0 and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and "shard" in cb["sample"]
    assert cb["shard_partial_stacks_relerr"] < 1e-12
    assert r["roofline"]["traffic"] is None and set(("bound", "achieved", "peak", "unit", "frac")) <= set(r["roofline"])
    assert len(r["per_rank_ms"]["ranks"]) == 2 and r["per_rank_ms"]["columns"] == ["stream", "exposed_collective", "finish"]
    assert r["config"]["traces_total"] == 128 and r["config"]["traces_per_gpu"] == 64
    assert r["value"] > 0 and r["scaling"] == "weak" and r["roofline"]["launches_per_call"] >= 1
    assert r["step_ms_gpu"]["n"] == 3 and r["step_ms_gpu"]["min"] <= r["step_ms_gpu"]["median"] <= r["step_ms_gpu"]["max"]
    r5 = _bench(["--gpus", "2", "--config", "cfg5", "--samples", "2048", "--kmax", "4", "--steps", "2", "--warmup", "1", "--no-cpu"], {"BENCH_BACKEND": "gloo"})
    assert r5["config"]["traces_per_gpu"] == 12500 and r5["config"]["traces_total"] == 25000 and "configs[4]" in r5["config"]["workload"]


@pytest.mark.gpu
@pytest.mark.parametrize("leg", ["split", "strong"])
def test_bench_prints_its_line_when_an_optional_leg_hangs(leg):
    """The N > 1 line must survive a schedule that hangs on real hardware (the overlapped schedules have never had a second RCCL rank): every
    optional leg runs under a deadline; when the last rank never shows up in a leg's collective, rank 0 prints the line measured so far --
    headline, roofline, per-rank columns, CPU baseline, the legs that did finish -- names the leg, and every rank leaves with status 0."""
    r = _bench(["--gpus", "2", "--traces", "64", "--samples", "4096", "--steps", "2", "--warmup", "1", "--strong-total", "200", "--leg-deadline", "6",
                "--inject-hang", leg], {"BENCH_BACKEND": "gloo"})
    assert r["aborted_leg"]["leg"] == ("schedule split" if leg == "split" else "strong scaling") and r["aborted_leg"]["deadline_s"] == 6
    assert r["value"] > 0 and r["n_gpus"] == 2 and "cpu_baseline" in r and "roofline" in r and len(r["per_rank_ms"]["ranks"]) == 2
    assert r["schedules"]["single"]["is_default"] and r["schedules"]["single"]["ms_per_step"] > 0
    if leg == "strong":
        assert set(r["schedules"]) == {"single", "split", "sharded-finish"} and "strong_scaling" not in r
    else:
        assert "split" not in r["schedules"]


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_contract_keys():
    """One GPU, small size: the line carries the contract keys, the roofline object, the per-step statistics, the D2H-inclusive
    figure and the CPU baseline with the full-size comparison against it."""
    r = _bench(["--traces", "200", "--samples", "8192", "--steps", "5", "--warmup", "2", "--no-extra"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "step_ms_gpu", "with_output_d2h"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["dtype"] == "f64" and r["vs_baseline"] is None and r["unit"] == "samples/s"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r["roofline"])
    assert r["cpu_baseline"]["rc"] == 0 and r["cpu_baseline"]["gpu_vs_cpu_relerr"]["tsPWS"] < 1e-5 and r["cpu_baseline"]["gpu_vs_cpu_relerr"]["ls"] < 1e-5
    assert r["step_ms_gpu"]["n"] == 5


@pytest.mark.parametrize("bad,field", [(2, "dt"), (5, "dt"), (3, "beg")])
def test_mismatching_sac_file_is_skipped_like_the_reference(tmp_path, bad, field):
    """The reference's reader (ts_pws1f.c:680-708) "skips" a trace whose dt is off by more than 1 % (or whose b is off by more than
    dt) by letting the next accepted trace overwrite its slot, and keeps the trace count at the FILE count (:708 updates a local):
    a skipped trace in the middle leaves a trailing all-zero row in the stack, a skipped LAST trace stays in its slot and IS stacked.
    The CLI reproduces that reader; the expected outputs are the oracle's tspws_main on the rows as the reference leaves them."""
    n, mtr, dt, beg = 3000, 6, 0.5, -10.0
    X = abi.synth_traces(mtr, n, seed=77)
    names = []
    for i in range(mtr):
        p = tmp_path / f"s{i}.sac"
        d, b = (dt * 1.05 if (i == bad and field == "dt") else dt), (beg + 3 * dt if (i == bad and field == "beg") else beg)
        abi.write_sac(str(p), X[i], d, b, year=2011, jday=10 + i)
        names.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = run_cli(tmp_path, "list.txt", "osac=skip", "TwoStage=3", "unbiased")
    assert f"skipping trace {bad}" in out and ("different dt" if field == "dt" else "different beg") in out
    rows = np.zeros((mtr, n), np.float32)          # calloc'd like the reference's sigall
    k = 0
    for i in range(mtr):
        rows[k] = X[i]                              # read into slot i - nskip ...
        if i != bad:
            k += 1                                  # ... and kept only when accepted
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(Kmax=3, unbiased=1), rows, dt=dt, beg=beg)
    ts, ls = abi.read_sac(tmp_path / "ts_pws_skip.sac"), abi.read_sac(tmp_path / "tl_skip.sac")
    assert abi.relerr(ts["data"], want["tsPWS"]) < 2e-6 and abi.relerr(ls["data"], want["ls"]) < 2e-6
    assert ts["f"][40] == float(mtr)                # user0 = the trace count the reference passes on: the file count


def test_batch_of_three_lists_equals_three_single_runs(sac_list, tmp_path):
    """`ts_pws @batch.txt [options]` (several ensembles in one process: one HIP start-up, the frame kept per (N, options), the files of
    ensemble i + 1 read while ensemble i is stacked): the outputs of every ensemble are BYTE for byte those of a single run with
    osac=<tag> (the reference's naming, ts_pws1f.c:335-349) -- also for the jackknife replicas of a two-stage run."""
    names = (sac_list / "list.txt").read_text().split()
    lists = {"pair_a": names, "pair_b": names[:20], "pair_c": names[7:]}
    for k, v in lists.items():
        (tmp_path / f"{k}.txt").write_text("\n".join(v) + "\n")
    (tmp_path / "batch.txt").write_text("pair_a.txt\npair_b.txt tagged_b\n# comment\npair_c.txt\n")
    tags = {"pair_a": "pair_a", "pair_b": "tagged_b", "pair_c": "pair_c"}
    for opts in (["rm"], ["TwoStage=4", "unbiased", "jackknife_n=4", "jackknife_d=1"]):
        single, batch = tmp_path / ("single" + str(len(opts))), tmp_path / ("batch" + str(len(opts)))
        single.mkdir(); batch.mkdir()
        for k in lists:
            run_cli(single, str(tmp_path / f"{k}.txt"), f"osac={tags[k]}", *opts)
        for k in lists:   # (the batch's lists are given relative to the working directory)
            (batch / f"{k}.txt").write_text((tmp_path / f"{k}.txt").read_text())
        (batch / "batch.txt").write_text((tmp_path / "batch.txt").read_text())
        run_cli(batch, "@batch.txt", *opts)
        made = sorted(f for f in os.listdir(single) if f.endswith(".sac"))
        assert len(made) == (2 + (8 if len(opts) > 1 else 0)) * 3, made
        for f in made:
            assert (single / f).read_bytes() == (batch / f).read_bytes(), f
