"""BASELINE configs[4] -- 100 000 traces x 131 072 samples, Morlet two-stage K = 10, trace-sharded 8 ways -- run through SURVEY
section 8(e)'s own parity plan on ONE MI355X (288 GB of HBM hold either ensemble):

  (1) 100 000 x 8 192 (8.2e8 samples: the largest trace count the reference can index -- its `itr * max` is 32-bit,
      /root/reference/src/ts_pws1f_lib.c:874, Tools/myallocs.c:35), sharded 8 ways, against the REFERENCE itself (oracle/_ref)
      when it was built, else the restatement: through tspws_hip_multi_stack with eight virtual shards of the one device, and
      through stack_sharded over eight gloo processes;
  (2) 100 000 x 131 072 generated on the device (52 GB): eight stack_local(first = r * 12 500, mtr_global = 100 000) shard
      passes summed + the finish stage, against the 64-bit restatement orc_tspws_main_mt on a host copy when the host has the
      memory, and always against direct FP64 group sums + the size-independent properties of the cfg3 tests.

The sum being sharded is partial_linear_stacks, ts_pws1f_lib.c:866-881: P[g] = sum of the traces with floor(i Kmax / mtr) == g."""
import ctypes as C
import importlib
import os
import socket

import numpy as np
import pytest

import abi
from test_comm_gpu import _multi_stack

pytestmark = pytest.mark.gpu
tspws = importlib.import_module("ts-pws_amd")
TOL32 = 2e-6          # float32 outputs against the CPU result (FP64 on both sides; north_star asks 1e-5)
KW = dict(Kmax=10, unbiased=1)
MTR = 100000


@pytest.fixture(scope="module")
def lib():
    return tspws.load()


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _call_nocopy(fn, params_in, Xh):
    """tspws_main-shaped call on host traces without copying them (no fold / rm here: the traces stay as they are)."""
    mtr, N = Xh.shape
    p = abi.t_tsPWS.from_buffer_copy(params_in)
    out = abi.t_tsPWS_out()
    ls, ts = np.zeros(N, np.float32), np.zeros(N, np.float32)
    fp = C.POINTER(C.c_float)
    out.ls, out.tsPWS = ls.ctypes.data_as(fp), ts.ctypes.data_as(fp)
    out.N, out.mtr = N, mtr
    d = abi.t_data()
    d.sigall = Xh.ctypes.data_as(fp)
    d.hdr.max, d.hdr.mtr, d.hdr.dt, d.hdr.beg = N, mtr, 1.0, 0.0
    rc = fn(C.byref(p), C.byref(out), C.byref(d))
    return rc, ls, ts


@pytest.fixture(scope="module")
def reduced(torch):
    """100 000 x 8 192 on the device + the CPU result of the whole ensemble (the reference when built)."""
    N = 8192
    X = tspws.synth(MTR, N, seed=5)
    Xh = X.cpu().numpy()
    ref = abi.ref()
    fn, kind = (ref.tspws_main, "reference") if ref is not None else (abi.oracle().orc_tspws_main_mt, "port")
    rc, ls, ts = _call_nocopy(fn, abi.default_params(**KW), Xh)
    assert rc == 0
    return dict(N=N, X=X, ls=ls, ts=ts, kind=kind)


def test_reduced_n_eight_virtual_shards_vs_reference(lib, torch, reduced):
    """SURVEY 8(e) parity plan (1) through the one-process path: tspws_hip_multi_stack, devices [0] * 8 -- eight 12 500-trace shards,
    group index from the GLOBAL trace index, rows of untouched groups zero, all-reduce, scale-sharded finish over eight shares."""
    ls, ts = _multi_stack(lib, torch, abi.default_params(**KW), reduced["X"], [0] * 8)
    assert abi.relerr(ls, reduced["ls"]) < TOL32 and abi.relerr(ts, reduced["ts"]) < TOL32, reduced["kind"]
    # ... and the one-shard engine on the same ensemble
    pl = tspws.Plan(tspws.resolve(abi.default_params(**KW), reduced["N"]), reduced["N"])
    ls1, ts1 = pl.stack_single(reduced["X"])
    torch.cuda.synchronize()
    assert abi.relerr(ls1.cpu().numpy(), reduced["ls"]) < TOL32 and abi.relerr(ts1.cpu().numpy(), reduced["ts"]) < TOL32
    assert abi.relerr(ls, ls1.cpu().numpy()) < 1e-6 and abi.relerr(ts, ts1.cpu().numpy()) < 1e-6


def _worker(rank, world, port, N, schedule, out_dir):
    import torch as th
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (one GPU here: RCCL wants one device per rank; gloo moves the same tensors)
    try:
        pl = tspws.Plan(tspws.resolve(abi.default_params(**KW), N), N)
        first, count = tspws.shard_range(MTR, rank, world)
        X = tspws.synth(count, N, seed=5, first=first)
        ls, ts = tspws.stack_sharded(pl, X, first, MTR, schedule=schedule)
        th.cuda.synchronize()
        if rank in (0, world - 1):
            np.save(os.path.join(out_dir, f"ls{rank}.npy"), ls.cpu().numpy())
            np.save(os.path.join(out_dir, f"ts{rank}.npy"), ts.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_reduced_n_eight_processes_over_gloo(lib, torch, reduced, tmp_path):
    """SURVEY 8(e) parity plan (1) through the one-process-per-GPU path: eight ranks (sharing this GPU, gloo as the collective) run
    stack_sharded on their 12 500-trace shards of the 100 000 -- the orchestration bench.py --gpus 8 --config cfg5 times."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(8, port, reduced["N"], "sharded-finish", str(tmp_path)), nprocs=8, join=True)
    for r in (0, 7):
        assert abi.relerr(np.load(tmp_path / f"ls{r}.npy"), reduced["ls"]) < TOL32, (r, reduced["kind"])
        assert abi.relerr(np.load(tmp_path / f"ts{r}.npy"), reduced["ts"]) < TOL32, (r, reduced["kind"])


def test_full_size_eight_shards_sum_and_finish(lib, torch):
    """SURVEY 8(e) parity plan (2): 100 000 x 131 072 (1.3e10 samples, 52 GB -- twelve times past the reference's 32-bit index).
    Eight shard passes with the global group index add up to the partial stacks of the whole ensemble; the finish stage on the
    sum is the stack.  Checked against direct FP64 sums of the traces, against the one-pass call, through exact properties, and --
    where the host has 80 GB to spare -- against the 64-bit restatement on a host copy of the same traces."""
    N, K, world = 131072, 10, 8
    p = tspws.resolve(abi.default_params(**KW), N)
    pl = tspws.Plan(p, N)
    X = tspws.synth(MTR, N, seed=8)
    assert X.numel() == 13107200000
    total = torch.zeros(K * N, dtype=torch.float64, device="cuda")
    for r in range(world):
        first, count = tspws.shard_range(MTR, r, world)
        assert count == 12500
        pl.stack_local(X[first:first + count], first, MTR)
        buf = pl.reduce_buffer(MTR)
        # a shard touches the groups its traces fall into (global index) and leaves zeros elsewhere
        rows = buf.view(K, N).abs().amax(dim=1).cpu().numpy() > 0
        want_rows = np.zeros(K, bool)
        want_rows[(first * K) // MTR:((first + count - 1) * K) // MTR + 1] = True
        np.testing.assert_array_equal(rows, want_rows)
        total += buf
    # direct FP64 group sums (group g = traces [10 000 g, 10 000 (g + 1)))
    for g in (0, 3, 9):
        direct = torch.zeros(N, dtype=torch.float64, device="cuda")
        for t0 in range(g * 10000, (g + 1) * 10000, 500):
            direct += X[t0:t0 + 500].double().sum(dim=0)
        assert float((total.view(K, N)[g] - direct).abs().max()) <= 1e-9, g
    # the sum of the shards IS the whole ensemble's buffer: finish on it and compare with the one-pass call
    pl.reduce_buffer(MTR).copy_(total)
    ls = torch.empty(N, dtype=torch.float32, device="cuda")
    ts = torch.empty(N, dtype=torch.float32, device="cuda")
    pl.stack_finish(MTR, ls, ts)
    ls1, ts1 = pl.stack_single(X)
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), ls1.cpu().numpy()) < 1e-6 and abi.relerr(ts.cpu().numpy(), ts1.cpu().numpy()) < 1e-6
    assert bool(torch.isfinite(ts).all()) and float(ts.abs().max()) > 0
    # the linear stack is the frame-filtered mean of all traces: ICWT(CWT(.)) of the direct mean (one transform pair of the engine itself)
    mean = total.view(K, N).sum(dim=0) / MTR
    Y = torch.zeros((1, pl.ncoef, 2), dtype=torch.float64, device="cuda")
    xr = torch.zeros((1, N), dtype=torch.float64, device="cuda")
    tspws.check(lib.tspws_hip_forward_f64(pl.h, mean.data_ptr(), 1, N, Y.data_ptr(), None), "forward")
    tspws.check(lib.tspws_hip_inverse(pl.h, Y.data_ptr(), 1, xr.data_ptr(), None), "inverse")
    torch.cuda.synchronize()
    assert abi.relerr(ls.cpu().numpy(), xr[0].cpu().numpy().astype(np.float32)) < 1e-5
    # host copy + the 64-bit restatement (the reference itself cannot index this ensemble)
    try:
        import psutil
        room = psutil.virtual_memory().available
    except Exception:
        room = 0
    if room > (80 << 30):
        Xh = np.empty((MTR, N), np.float32)
        step = 5000
        for t0 in range(0, MTR, step):
            Xh[t0:t0 + step] = X[t0:t0 + step].cpu().numpy()
        rc, ls_h, ts_h = _call_nocopy(abi.oracle().orc_tspws_main_mt, abi.default_params(**KW), Xh)
        del Xh
        assert rc == 0
        assert abi.relerr(ls.cpu().numpy(), ls_h) < TOL32 and abi.relerr(ts.cpu().numpy(), ts_h) < TOL32
    else:
        print(f"(host has {room >> 30} GB available: the 52-GB host copy for orc_tspws_main_mt was skipped)")


@pytest.mark.parametrize("schedule", ["single", "split", "sharded-finish"])
def test_three_schedules_over_gloo(lib, torch, tmp_path, schedule):
    """bench.py --schedule: the three placements of the one logical reduction (stack_sharded) with the real engine, three ranks on
    this GPU over gloo.  "single" (one all-reduce, redundant finish) and "split" (pieces, redundant finish in pieces) keep the
    one-GPU accumulation order: with shards whose sums are exact -- integer-valued traces -- they match the one-GPU outputs bit for
    bit; "sharded-finish" adds its partial reconstructions in the collective's order (1e-6)."""
    import torch.multiprocessing as mp
    N, mtr, world = 8192, 300, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_int_worker, args=(world, port, N, mtr, schedule, str(tmp_path)), nprocs=world, join=True)
    pl = tspws.Plan(tspws.resolve(abi.default_params(**KW), N), N)
    X = (tspws.synth(mtr, N, seed=17) * 64).round()
    ls, ts = pl.stack_single(X)
    torch.cuda.synchronize()
    for r in range(world):
        a, b = np.load(tmp_path / f"ls{r}.npy"), np.load(tmp_path / f"ts{r}.npy")
        if schedule == "sharded-finish":
            assert abi.relerr(a, ls.cpu().numpy()) < 1e-6 and abi.relerr(b, ts.cpu().numpy()) < 1e-6
        else:
            np.testing.assert_array_equal(a, ls.cpu().numpy())
            np.testing.assert_array_equal(b, ts.cpu().numpy())


def _int_worker(rank, world, port, N, mtr, schedule, out_dir):
    import torch as th
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pl = tspws.Plan(tspws.resolve(abi.default_params(**KW), N), N)
        first, count = tspws.shard_range(mtr, rank, world)
        X = (tspws.synth(count, N, seed=17, first=first) * 64).round()   # small integers: every partial sum is exact in FP64
        ls, ts = tspws.stack_sharded(pl, X, first, mtr, schedule=schedule)
        th.cuda.synchronize()
        np.save(os.path.join(out_dir, f"ls{rank}.npy"), ls.cpu().numpy())
        np.save(os.path.join(out_dir, f"ts{rank}.npy"), ts.cpu().numpy())
    finally:
        dist.destroy_process_group()
