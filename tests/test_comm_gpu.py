"""Several devices of one process (csrc/comm.hip): the all-reduce behind the C ABI and the sharded whole call that the
drop-in tspws_main uses under TSPWS_DEVICES.  A one-GPU box can exercise (a) RCCL itself with a one-device communicator
(ncclCommInitAll(1), TSPWS_COMM=rccl) and (b) the N-way bookkeeping -- shard ranges from the global trace index, zero rows
of untouched groups, scale-sharded finish, replica blocks -- by naming the one device several times, which selects the
library's own reduction kernel ("local" backend; RCCL refuses duplicate devices).  Everything is compared with the
single-device engine and with the oracle's tspws_main (the reference sums the traces of one host array,
/root/reference/src/ts_pws1f_lib.c:866-881)."""
import ctypes as C
import importlib

import numpy as np
import pytest

import abi

pytestmark = pytest.mark.gpu
tspws = importlib.import_module("ts-pws_amd")
TOL32 = 2e-6


@pytest.fixture(scope="module")
def lib():
    return tspws.load()


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _comm(lib, devs):
    h = C.c_void_p()
    arr = (C.c_int * len(devs))(*devs)
    tspws.check(lib.tspws_hip_comm_create(C.byref(h), len(devs), arr), "comm_create")
    return h


def test_allreduce_through_rccl_on_one_device(lib, torch, monkeypatch):
    monkeypatch.setenv("TSPWS_COMM", "rccl")
    h = _comm(lib, [0])
    try:
        assert lib.tspws_hip_comm_size(h) == 1 and lib.tspws_hip_comm_device(h, 0) == 0
        assert lib.tspws_hip_comm_backend(h).decode().startswith("rccl ")
        x = torch.arange(100000, dtype=torch.float64, device="cuda") * 0.5
        want = x.clone()
        ptrs = (C.c_void_p * 1)(x.data_ptr())
        torch.cuda.synchronize()
        tspws.check(lib.tspws_hip_allreduce_f64(h, ptrs, x.numel(), None), "allreduce")
        tspws.check(lib.tspws_hip_sync(lib.tspws_hip_comm_stream(h, 0)), "sync")
        assert torch.equal(x, want)   # the sum over one rank
    finally:
        lib.tspws_hip_comm_destroy(h)


def test_local_backend_adds_in_list_order(lib, torch):
    h = _comm(lib, [0, 0, 0])
    try:
        assert lib.tspws_hip_comm_backend(h).decode() == "local" and lib.tspws_hip_comm_size(h) == 3
        rng = np.random.default_rng(3)
        host = [rng.standard_normal(70001) for _ in range(3)]
        bufs = [torch.as_tensor(a, device="cuda") for a in host]
        ptrs = (C.c_void_p * 3)(*[b.data_ptr() for b in bufs])
        torch.cuda.synchronize()
        tspws.check(lib.tspws_hip_allreduce_f64(h, ptrs, 70001, None), "allreduce")
        for i in range(3):
            tspws.check(lib.tspws_hip_sync(lib.tspws_hip_comm_stream(h, i)), "sync")
        want = (host[0] + host[1]) + host[2]
        for b in bufs:
            np.testing.assert_array_equal(b.cpu().numpy(), want)
    finally:
        lib.tspws_hip_comm_destroy(h)


def _multi_stack(lib, torch, params, X, devs, sel=None):
    """tspws_hip_multi_stack(_jackknife) on shards of the device tensor X; returns numpy outputs."""
    mtr, N = X.shape
    p = tspws.resolve(params, N)
    m = C.c_void_p()
    arr = (C.c_int * len(devs))(*devs)
    tspws.check(lib.tspws_hip_multi_create(C.byref(m), len(devs), arr, p.type, p.J, p.V, N, p.s0, p.b0, p.w0, int(p.uni)), "multi_create")
    try:
        shards = []
        for r in range(len(devs)):
            f, c = tspws.shard_range(mtr, r, len(devs))
            shards.append(X[f:f + c].contiguous() if c else None)
        ptrs = (C.c_void_p * len(devs))(*[s.data_ptr() if s is not None else None for s in shards])
        ls = torch.empty(N, dtype=torch.float32, device="cuda")
        ts = torch.empty(N, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if sel is None:
            tspws.check(lib.tspws_hip_multi_stack(m, C.byref(p), ptrs, N, mtr, ls.data_ptr(), ts.data_ptr()), "multi_stack")
            return ls.cpu().numpy(), ts.cpu().numpy()
        Cn = sel.shape[0]
        jl, jt, jm = np.zeros((Cn, N), np.float32), np.zeros((Cn, N), np.float32), np.zeros(Cn, np.uint32)
        tspws.check(lib.tspws_hip_multi_stack_jackknife(m, C.byref(p), ptrs, N, mtr, ls.data_ptr(), ts.data_ptr(), sel.ctypes.data, Cn,
                                                       jl.ctypes.data, jt.ctypes.data, jm.ctypes.data), "multi_stack_jackknife")
        return ls.cpu().numpy(), ts.cpu().numpy(), jl, jt, jm
    finally:
        lib.tspws_hip_multi_destroy(m)


@pytest.mark.parametrize("kw,mtr,N,devs", [
    (dict(Kmax=10, unbiased=1), 100, 4096, [0, 0, 0]),       # scale-sharded finish over three shares
    (dict(Kmax=7), 53, 3001, [0, 0]),                          # odd N: no sharded finish, the first device finishes alone
    (dict(), 40, 2048, [0, 0, 0, 0]),                          # single-stage: ST || PS are what the devices add
    (dict(type=-3, Kmax=4, wu=1.0), 30, 4096, [0, 0, 0]),
    (dict(Kmax=2, unbiased=1), 2, 2048, [0, 0, 0]),            # fewer traces than devices: empty shards add zeros
    (dict(Kmax=10, unbiased=1), 64, 8192, [0] * 8),            # eight shares of the scales
])
def test_sharded_call_matches_the_single_device_engine(lib, torch, kw, mtr, N, devs):
    X = tspws.synth(mtr, N, seed=61)
    ls, ts = _multi_stack(lib, torch, abi.default_params(**kw), X, devs)
    pl = tspws.Plan(tspws.resolve(abi.default_params(**kw), N), N)
    ls0, ts0 = pl.stack_single(X)
    torch.cuda.synchronize()
    assert abi.relerr(ls, ls0.cpu().numpy()) < 1e-6 and abi.relerr(ts, ts0.cpu().numpy()) < 1e-6
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X.cpu().numpy())
    assert abi.relerr(ls, want["ls"]) < TOL32 and abi.relerr(ts, want["tsPWS"]) < TOL32


def test_sharded_jackknife_matches_the_oracle(lib, torch):
    mtr, N, K, n = 240, 4096, 6, 5
    kw = dict(Kmax=K, unbiased=1, jackknife_n=n, jackknife_d=2)
    Cn = abi.binomial(n, 2)
    X = tspws.synth(mtr, N, seed=62)
    rng = np.random.default_rng(9)
    times = (1262304000 + 86400 * np.sort(rng.integers(0, 2 * 365, mtr))).astype(np.int64)
    sel = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 2, n, Cn) == 0
    ls, ts, jl, jt, jm = _multi_stack(lib, torch, abi.default_params(**kw), X, [0, 0, 0], sel)   # replica blocks of 4 | 3 | 3
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X.cpu().numpy(), times=times)
    assert abi.relerr(ls, want["ls"]) < TOL32 and abi.relerr(ts, want["tsPWS"]) < TOL32
    np.testing.assert_array_equal(jm, want["jk_mtr"])
    for c in range(Cn):
        assert abi.relerr(jt[c], want["jk_ts"][c]) < TOL32 and abi.relerr(jl[c], want["jk_ls"][c]) < TOL32, c


@pytest.mark.parametrize("devices,comm", [("0,0,0", None), ("0", "rccl"), ("all", "rccl")])
def test_tspws_main_over_a_device_list(lib, monkeypatch, devices, comm):
    """The drop-in itself under TSPWS_DEVICES: host buffers in, shards uploaded per device, prologue mirrored back, outputs and
    replicas out -- against the oracle's tspws_main, return codes and mutated traces included."""
    monkeypatch.setenv("TSPWS_DEVICES", devices)
    if comm:
        monkeypatch.setenv("TSPWS_COMM", comm)
    try:
        X = abi.synth_traces(90, 2048, seed=71)
        for kw, beg in ((dict(Kmax=5, unbiased=1), 0.0), (dict(), 0.0), (dict(Kmax=4, lrm=1, fold=1), -0.5 * 2047), (dict(type=-3, lrm=1), 0.0)):
            got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X, beg=beg)
            want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X, beg=beg)
            assert got["rc"] == want["rc"] == 0
            assert abi.relerr(got["sigall"], want["sigall"]) < 1e-6
            assert abi.relerr(got["ls"], want["ls"]) < TOL32 and abi.relerr(got["tsPWS"], want["tsPWS"]) < TOL32, kw
            assert got["params"].fold == want["params"].fold and got["params"].J == want["params"].J
        times = (1262304000 + 86400 * 3 * np.arange(90)).astype(np.int64)
        kw = dict(Kmax=5, unbiased=1, jackknife_n=4, jackknife_d=1)
        got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X, times=times)
        want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X, times=times)
        assert got["rc"] == 0 and abi.relerr(got["tsPWS"], want["tsPWS"]) < TOL32
        np.testing.assert_array_equal(got["jk_mtr"], want["jk_mtr"])
        assert max(abi.relerr(got["jk_ts"][c], want["jk_ts"][c]) for c in range(4)) < TOL32
        assert max(abi.relerr(got["jk_ls"][c], want["jk_ls"][c]) for c in range(4)) < TOL32
        # convergence curves need the whole ensemble on one device: the call takes the single-device path and still answers
        kw = dict(convergence=1, Kmax=3)
        got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X[:12])
        want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X[:12])
        assert got["rc"] == 0 and abi.relerr(got["conv_tsPWS_sim"], want["conv_tsPWS_sim"]) < 1e-6
    finally:
        lib.tspws_main_release()


def test_torch_nccl_backend_reduces_the_plan_buffers(lib, torch):
    """bench.py --gpus N and stack_sharded() hand views of the library's own device buffers (partial stacks, replica rows) to
    torch.distributed with backend "nccl" (= RCCL).  One rank is all a 1-GPU box can start, but it still goes through RCCL's
    registration of the buffer, the async work handle on the collective's stream and the stream hand-over back to the caller."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        N, mtr, K = 8192, 120, 6
        p = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
        pl = tspws.Plan(p, N)
        X = tspws.synth(mtr, N, seed=4)
        ls0, ts0 = pl.stack(X)
        ls0, ts0 = ls0.clone(), ts0.clone()
        buf = pl.reduce_buffer(mtr).view(K, N)
        half = K // 2
        pl.partial_stacks_range(X, 0, mtr, 0, half)
        w1 = dist.all_reduce(buf[:half], op=dist.ReduceOp.SUM, async_op=True)
        pl.partial_stacks_range(X, 0, mtr, half, K)
        w2 = dist.all_reduce(buf[half:], op=dist.ReduceOp.SUM, async_op=True)
        w1.wait(); w2.wait()
        ls = torch.empty(N, dtype=torch.float32, device="cuda")
        ts = torch.empty(N, dtype=torch.float32, device="cuda")
        pl.stack_finish(mtr, ls, ts)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(ls.cpu().numpy(), ls0.cpu().numpy())
        np.testing.assert_array_equal(ts.cpu().numpy(), ts0.cpu().numpy())
        x2 = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
        dist.all_reduce(x2, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_device_list_call_fails_after_the_prologue_like_the_reference(lib, monkeypatch):
    """A frame that cannot be built (J resolves to 0) fails with 4 AFTER fold / mean removal have rewritten the traces
    (/root/reference/src/ts_pws1f_lib.c:71-88, :159-169, :199-204); the several-device call must leave the same traces behind as the
    single-device call and the oracle."""
    kw = dict(type=-3, s0=4.823433067845736, fmin=0.04796741189352445, wu=1.5, Kmax=12, lrm=1)
    X = abi.synth_traces(9, 1000, seed=3) + np.float32(0.25)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X)
    assert want["rc"] == 4 and want["params"].J == 0
    monkeypatch.setenv("TSPWS_DEVICES", "0,0")
    try:
        got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
        assert got["rc"] == 4
        assert abi.relerr(got["sigall"], want["sigall"]) < 1e-6 and not np.array_equal(got["sigall"], X)
        assert not got["ls"].any() and not got["tsPWS"].any()
    finally:
        lib.tspws_main_release()


def test_reduce_to_owner_on_the_local_backend(lib, torch):
    """tspws_hip_reduce_f64: the sum lands on the root only (list order), the other buffers keep their contents -- what the sharded
    jackknife uses for the replica rows of a block's owner."""
    h = C.c_void_p()
    devs = (C.c_int * 3)(0, 0, 0)
    tspws.check(lib.tspws_hip_comm_create(C.byref(h), 3, devs), "comm_create")
    try:
        rng = np.random.default_rng(4)
        host = [rng.standard_normal(50003) for _ in range(3)]
        for root in (0, 2):
            bufs = [torch.as_tensor(a, device="cuda") for a in host]
            ptrs = (C.c_void_p * 3)(*[b.data_ptr() for b in bufs])
            torch.cuda.synchronize()
            tspws.check(lib.tspws_hip_reduce_f64(h, ptrs, 50003, root, None), "reduce")
            for i in range(3):
                tspws.check(lib.tspws_hip_sync(lib.tspws_hip_comm_stream(h, i)), "sync")
            want = (host[0] + host[1]) + host[2]
            for i, b in enumerate(bufs):
                np.testing.assert_array_equal(b.cpu().numpy(), want if i == root else host[i])
    finally:
        lib.tspws_hip_comm_destroy(h)


@pytest.mark.parametrize("kw,mtr,N,ndev", [(dict(Kmax=10, unbiased=1), 96, 8192, 8), (dict(Kmax=7), 50, 4096, 3), (dict(Kmax=10, unbiased=1), 37, 3001, 4)])
def test_c_schedules_of_the_device_list_call(lib, torch, monkeypatch, kw, mtr, N, ndev):
    """tspws_hip_multi_stack places the one logical reduction by TSPWS_SCHEDULE like ts-pws_amd.stack_sharded: `single` and `split`
    (pieces on the communicator's second streams, finish in pieces on the first device) are bit-identical on integer-valued traces
    -- every partial sum is exact --, `sharded-finish` ends its sum over octaves in another order (1e-6).  Virtual shards of the one GPU."""
    X = torch.round(tspws.synth(mtr, N, seed=63) * 64.0).contiguous()     # integers / 64: FP64 sums are exact in any order
    out = {}
    for sch in ("single", "split", "sharded-finish"):
        monkeypatch.setenv("TSPWS_SCHEDULE", sch)
        out[sch] = _multi_stack(lib, torch, abi.default_params(**kw), X, [0] * ndev)
    monkeypatch.setenv("TSPWS_SCHEDULE", "nonsense")
    with pytest.raises(tspws.TspwsError):
        _multi_stack(lib, torch, abi.default_params(**kw), X, [0] * ndev)
    monkeypatch.delenv("TSPWS_SCHEDULE")
    np.testing.assert_array_equal(out["single"][0], out["split"][0])
    np.testing.assert_array_equal(out["single"][1], out["split"][1])
    assert abi.relerr(out["sharded-finish"][0], out["single"][0]) < 1e-6 and abi.relerr(out["sharded-finish"][1], out["single"][1]) < 1e-6
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), X.cpu().numpy())
    assert abi.relerr(out["single"][0], want["ls"]) < TOL32 and abi.relerr(out["single"][1], want["tsPWS"]) < TOL32


def test_upload_over_a_device_list_in_pieces(lib, torch):
    """tspws_hip_multi_upload pins the host array in page-aligned 128-MB pieces cut over the WHOLE array and lets every device copy
    its shard as soon as its first piece is pinned: shards that start and end in the middle of pieces (and of pages) arrive byte for byte."""
    N, mtr, ndev = 8192 + 4, 11003, 3                       # 361 MB: three pieces; shard boundaries inside pieces, rows not page-sized
    p = tspws.resolve(abi.default_params(Kmax=4), 8192)
    m = C.c_void_p()
    arr = (C.c_int * ndev)(*([0] * ndev))
    tspws.check(lib.tspws_hip_multi_create(C.byref(m), ndev, arr, p.type, p.J, p.V, 8192, p.s0, p.b0, p.w0, int(p.uni)), "multi_create")
    try:
        raw = np.random.default_rng(8).integers(0, 1 << 30, mtr * N + 1024 + 7, dtype=np.int32).view(np.float32)
        off = ((-raw.ctypes.data) % 4096) // 4 + 5          # 20 bytes into a page
        H = raw[off:off + mtr * N].reshape(mtr, N)
        sh, dl, dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        tspws.check(lib.tspws_hip_multi_upload(m, H.ctypes.data, N, mtr, C.byref(sh), C.byref(dl), C.byref(dt)), "multi_upload")
        shards = C.cast(sh, C.POINTER(C.c_void_p))
        for r in range(ndev):
            f, c = tspws.shard_range(mtr, r, ndev)
            got = np.empty((c, N), np.float32)
            assert lib.tspws_hip_download(C.c_void_p(got.ctypes.data), C.c_void_p(shards[r]), C.c_size_t(got.nbytes), None) == 0
            assert np.array_equal(got.view(np.int32), H[f:f + c].view(np.int32)), r
    finally:
        lib.tspws_hip_multi_destroy(m)


def test_device_list_request_runs_on_the_first_listed_device(lib, monkeypatch):
    """Random subsampling under TSPWS_DEVICES: one device only, and it is the first of the list -- "7,7" has no such device on this box
    and must fail (5) instead of running on device 0; "0,0" gives the single-device result."""
    X = abi.synth_traces(12, 1024, seed=5)
    kw = dict(subsmpl_N=2, subsmpl_p=0.5)
    abi.srand(3)
    want = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
    monkeypatch.setenv("TSPWS_DEVICES", "0,0")
    abi.srand(3)
    got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
    assert got["rc"] == 0 and np.array_equal(got["sub_ts"], want["sub_ts"]) and np.array_equal(got["tsPWS"], want["tsPWS"])
    if lib.tspws_hip_device_count() <= 7:
        monkeypatch.setenv("TSPWS_DEVICES", "7,7")
        assert abi.run_main(lib.tspws_main, abi.default_params(**kw), X)["rc"] == 5


@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_physical_devices_all_schedules_and_the_jackknife(lib, monkeypatch, ndev):
    """The first run with SEVERAL physical GPUs (skipped on a one-GPU box -- every test above stands in with virtual shards or one RCCL
    rank): the drop-in tspws_main under TSPWS_DEVICES over real RCCL ranks with each TSPWS_SCHEDULE -- `split` / `sharded-finish` issue
    reductions on the communicator from two streams per device and ncclReduce to owners --, then the sharded jackknife (replica rows reduced
    to their owners), each against the one-device call on the same host traces."""
    if lib.tspws_hip_device_count() < ndev:
        pytest.skip(f"needs {ndev} GPUs")
    mtr, N = 240, 8192
    X = np.round(abi.synth_traces(mtr, N, seed=91) * 64.0).astype(np.float32)   # integers / 64: shard sums are exact in any order
    devs = ",".join(str(i) for i in range(ndev))
    try:
        for kw in (dict(Kmax=10, unbiased=1), dict()):
            monkeypatch.delenv("TSPWS_DEVICES", raising=False)
            monkeypatch.delenv("TSPWS_SCHEDULE", raising=False)
            want = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
            assert want["rc"] == 0
            monkeypatch.setenv("TSPWS_DEVICES", devs)
            for sch in ("single", "split", "sharded-finish"):
                monkeypatch.setenv("TSPWS_SCHEDULE", sch)
                got = abi.run_main(lib.tspws_main, abi.default_params(**kw), X)
                assert got["rc"] == 0, (kw, sch)
                if sch != "sharded-finish" and kw:
                    np.testing.assert_array_equal(got["ls"], want["ls"]); np.testing.assert_array_equal(got["tsPWS"], want["tsPWS"])
                else:
                    assert abi.relerr(got["ls"], want["ls"]) < 1e-6 and abi.relerr(got["tsPWS"], want["tsPWS"]) < 1e-6, (kw, sch)
        monkeypatch.delenv("TSPWS_SCHEDULE", raising=False)
        times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)
        pj = dict(Kmax=6, unbiased=1, jackknife_n=5, jackknife_d=1)
        monkeypatch.delenv("TSPWS_DEVICES", raising=False)
        want = abi.run_main(lib.tspws_main, abi.default_params(**pj), X, times=times)
        monkeypatch.setenv("TSPWS_DEVICES", devs)
        got = abi.run_main(lib.tspws_main, abi.default_params(**pj), X, times=times)
        assert got["rc"] == 0 and want["rc"] == 0
        np.testing.assert_array_equal(got["jk_mtr"], want["jk_mtr"])
        assert abi.relerr(got["tsPWS"], want["tsPWS"]) < 1e-6 and abi.relerr(got["jk_ts"], want["jk_ts"]) < 1e-6 and abi.relerr(got["jk_ls"], want["jk_ls"]) < 1e-6
    finally:
        lib.tspws_main_release()
