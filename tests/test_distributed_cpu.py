"""world_size-2 gloo test of the trace-sharded path on CPU.

The product's multi-GPU orchestration (ts-pws_amd.stack_sharded: shard-local half -> ONE all-reduce ->
finish) is exercised with an oracle-backed stand-in for the device plan (tests may use the oracle as the
checker); the sharded result must equal the unsharded oracle call."""
import ctypes as C
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import abi

tspws = importlib.import_module("ts-pws_amd")


class OraclePlan:
    """CPU stand-in with the interface stack_sharded needs; arithmetic by the oracle."""

    def __init__(self, params, N, ranged=True):
        self.p = abi.resolve(params, N)
        self.params = self.p  # stack_sharded looks at params.Kmax to choose the overlapped two-half reduction
        if not ranged:
            self.partial_stacks_range = None
        self.N = N
        self.f = abi.OracleFrame.from_params(self.p, N)
        self.buf = None

    def two_stage(self, mtr_global):
        return bool(self.p.Kmax) and self.p.Kmax <= mtr_global

    def reduce_buffer(self, mtr_global):
        n = self.p.Kmax * self.N if self.two_stage(mtr_global) else 4 * self.f.ncoef
        if self.buf is None or self.buf.numel() != n:
            self.buf = torch.zeros(n, dtype=torch.float64)
        return self.buf

    def stack_local(self, traces, first, mtr_global):
        x = traces.numpy()
        buf = self.reduce_buffer(mtr_global)
        buf.zero_()
        if self.two_stage(mtr_global):
            P = buf.numpy().reshape(self.p.Kmax, self.N)
            for i in range(x.shape[0]):
                g = int(np.floor(float((first + i) * self.p.Kmax) / float(mtr_global)))  # ts_pws1f_lib.c:876, GLOBAL index
                P[g] += x[i].astype(np.float64)
        else:
            nc = self.f.ncoef
            ST = buf.numpy()[:2 * nc].view(np.complex128)
            PS = buf.numpy()[2 * nc:].view(np.complex128)
            for i in range(x.shape[0]):
                Y = self.f.forward(x[i].astype(np.float64))
                abi.oracle().orc_accumulate(ST.ctypes.data, PS.ctypes.data, Y.ctypes.data, nc)

    def partial_stacks_range(self, traces, first, mtr_global, g_begin, g_end):
        x = traces.numpy()
        P = self.reduce_buffer(mtr_global).numpy().reshape(self.p.Kmax, self.N)
        P[g_begin:g_end] = 0
        for i in range(x.shape[0]):
            g = int(np.floor(float((first + i) * self.p.Kmax) / float(mtr_global)))
            if g_begin <= g < g_end:
                P[g] += x[i].astype(np.float64)

    def stack_finish(self, mtr_global, ls, ts):
        nc = self.f.ncoef
        orc = abi.oracle()
        if self.two_stage(mtr_global):
            K = self.p.Kmax
            P = self.buf.numpy().reshape(K, self.N)
            ST = np.zeros(nc, np.complex128)
            PS = np.zeros(nc, np.complex128)
            for g in range(K):
                Y = self.f.forward(P[g])
                orc.orc_accumulate(ST.ctypes.data, PS.ctypes.data, Y.ctypes.data, nc)
        else:
            K = mtr_global
            ST = self.buf.numpy()[:2 * nc].view(np.complex128).copy()
            PS = self.buf.numpy()[2 * nc:].view(np.complex128).copy()
        OUT = np.zeros(nc, np.complex128)
        orc.orc_weight(OUT.ctypes.data, ST.ctypes.data, PS.ctypes.data, nc, K, mtr_global, self.p.wu, self.p.unbiased)
        ts.copy_(torch.from_numpy(self.f.inverse(OUT).astype(np.float32)))
        ls.copy_(torch.from_numpy(self.f.inverse(ST).astype(np.float32) / np.float32(mtr_global)))


    # ---- scale-sharded finish: same interface as ts-pws_amd.Plan's finish_shard / stack_finish_scales / epilogue ----
    def finish_shard(self, mtr_global, rank, world):
        if not self.two_stage(mtr_global):
            return None
        S = self.f.S
        D = self.f.D
        octs = [0] + [s for s in range(1, S) if D[s] != D[s - 1]] + [S]   # decimation octaves
        n = len(octs) - 1
        lo, hi = rank * n // world, (rank + 1) * n // world               # any contiguous split of whole octaves is valid
        return (octs[lo], octs[hi]) if lo < hi else (0, 0)

    def stack_finish_scales(self, mtr_global, s_begin, s_end, x2):
        nc, K, N = self.f.ncoef, self.p.Kmax, self.N
        orc = abi.oracle()
        P = self.buf.numpy().reshape(K, N)
        ST = np.zeros(nc, np.complex128)
        PS = np.zeros(nc, np.complex128)
        for g in range(K):
            Y = self.f.forward(P[g])
            orc.orc_accumulate(ST.ctypes.data, PS.ctypes.data, Y.ctypes.data, nc)
        OUT = np.zeros(nc, np.complex128)
        orc.orc_weight(OUT.ctypes.data, ST.ctypes.data, PS.ctypes.data, nc, K, mtr_global, self.p.wu, self.p.unbiased)
        off = np.concatenate(([0], np.cumsum(self.f.Ns.astype(np.int64))))  # ragged container: d[s+1] = d[s] + N[s]
        mask = np.zeros(nc, bool)
        mask[off[s_begin]:off[s_end]] = True                                # the reconstruction is a sum over scales
        x2.numpy()[:N] = self.f.inverse(np.where(mask, OUT, 0))
        x2.numpy()[N:] = self.f.inverse(np.where(mask, ST, 0))

    def epilogue(self, x2, mtr_global, ls, ts):
        N = self.N
        ts.copy_(torch.from_numpy(x2.numpy()[:N].astype(np.float32)))
        ls.copy_(torch.from_numpy(x2.numpy()[N:].astype(np.float32) / np.float32(mtr_global)))

    # ---- trace-sharded jackknife: same interface as ts-pws_amd.Plan's jackknife_* methods ----
    def jackknife_buffer(self, Cn):
        n = Cn * self.p.Kmax * self.N
        if getattr(self, "jbuf", None) is None or self.jbuf.numel() != n:
            self.jbuf = torch.zeros(n, dtype=torch.float64)
        return self.jbuf

    def jackknife_local(self, traces, first, mtr_global, sel):
        x = traces.numpy()
        K, N, Cn = self.p.Kmax, self.N, sel.shape[0]
        self.stack_local(traces, first, mtr_global)  # the plain groups
        rows = self.jackknife_buffer(Cn)
        rows.zero_()
        R = rows.numpy().reshape(Cn, K, N)
        for c in range(Cn):
            Kc = int((sel[c] == 1).sum())
            k = int((sel[c, :first] == 1).sum())     # rank among ALL selected traces (ts_pws1f_lib.c:766): global prefix
            for i in range(x.shape[0]):
                if sel[c, first + i] != 1:
                    continue
                g = int(np.floor(float(k * K) / float(Kc)))
                R[c, g] += x[i].astype(np.float64)
                k += 1

    def jackknife_finish(self, mtr_global, sel, c_begin, c_end, ls_out, ts_out, mtr_out):
        K, N, Cn, nc = self.p.Kmax, self.N, sel.shape[0], self.f.ncoef
        orc = abi.oracle()
        R = self.jackknife_buffer(Cn).numpy().reshape(Cn, K, N)
        for c in range(c_begin, c_end):
            Kc = int((sel[c] == 1).sum())
            ST = np.zeros(nc, np.complex128)
            PS = np.zeros(nc, np.complex128)
            for g in range(K):
                Y = self.f.forward(R[c, g])
                orc.orc_accumulate(ST.ctypes.data, PS.ctypes.data, Y.ctypes.data, nc)
            OUT = np.zeros(nc, np.complex128)
            orc.orc_weight(OUT.ctypes.data, ST.ctypes.data, PS.ctypes.data, nc, K, Kc, self.p.wu, self.p.unbiased)
            ts_out[c].copy_(torch.from_numpy(self.f.inverse(OUT).astype(np.float32)))
            ls_out[c].copy_(torch.from_numpy((R[c].sum(axis=0) * (1.0 / Kc)).astype(np.float32)))
            mtr_out[c] = Kc


def _jk_times(mtr):
    rng = np.random.default_rng(12)
    return (1262304000 + 86400 * np.sort(rng.integers(0, 2 * 365, mtr))).astype(np.int64)


def _jk_worker(rank, world, port, kw, mtr, N, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = abi.synth_traces(mtr, N, seed=17)
        p = abi.default_params(**kw)
        Cn = abi.binomial(p.jackknife_n, p.jackknife_d)
        sel = np.zeros((Cn, mtr), np.int8)
        assert abi.oracle().orc_jackknife_plan(sel.ctypes.data, _jk_times(mtr).ctypes.data, mtr, p.jackknife_d, p.jackknife_n, Cn) == 0
        first, count = tspws.shard_range(mtr, rank, world)
        plan = OraclePlan(p, N)
        ls, ts, jl, jt, jm = tspws.jackknife_sharded(plan, torch.from_numpy(X[first:first + count]), sel, first, mtr)
        np.savez(os.path.join(out_dir, f"jk{rank}.npz"), ls=ls.numpy(), ts=ts.numpy(), jl=jl.numpy(), jt=jt.numpy(), jm=jm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw,mtr,world", [
    (dict(Kmax=3, unbiased=1, jackknife_n=4, jackknife_d=1), 37, 2),   # 4 replicas on 2 ranks: two each
    (dict(Kmax=2, jackknife_n=3, jackknife_d=1), 19, 3),               # one replica per rank
    (dict(Kmax=2, jackknife_n=4, jackknife_d=2), 21, 2),               # 6 replicas, deletion pairs
])
def test_sharded_jackknife_matches_unsharded(tmp_path, kw, mtr, world):
    """jackknife_sharded over gloo: shard-local rows from GLOBAL ranks, per-replica reductions to the owners, owners finish,
    one all-reduce of the outputs -- every rank ends with the oracle's replicas and main stack."""
    N = 1024
    mp.spawn(_jk_worker, args=(world, _free_port(), kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), abi.synth_traces(mtr, N, seed=17), times=_jk_times(mtr))
    assert want["jk_mtr"].all()
    for r in range(world):
        got = np.load(tmp_path / f"jk{r}.npz")
        np.testing.assert_array_equal(got["jm"], want["jk_mtr"])
        assert abi.relerr(got["ls"], want["ls"]) < 2e-6 and abi.relerr(got["ts"], want["tsPWS"]) < 2e-6
        for c in range(len(want["jk_mtr"])):
            assert abi.relerr(got["jl"][c], want["jk_ls"][c]) < 2e-6
            assert abi.relerr(got["jt"][c], want["jk_ts"][c]) < 2e-6


def _worker(rank, world, port, kw, mtr, N, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = abi.synth_traces(mtr, N, seed=17)
        first, count = tspws.shard_range(mtr, rank, world)
        plan = OraclePlan(abi.default_params(**kw), N)
        ls, ts = tspws.stack_sharded(plan, torch.from_numpy(X[first:first + count]), first, mtr, schedule=os.environ.get("TEST_SCHEDULE") or None)
        np.save(os.path.join(out_dir, f"ls{rank}.npy"), ls.numpy())
        np.save(os.path.join(out_dir, f"ts{rank}.npy"), ts.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kw", [dict(Kmax=5, unbiased=1), dict(), dict(Kmax=3, type=-3)])
def test_two_rank_shards_match_unsharded(tmp_path, kw):
    mtr, N, world = 23, 1024, 2  # odd trace count: shards of 11 and 12, group boundaries inside shards
    mp.spawn(_worker, args=(world, _free_port(), kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), abi.synth_traces(mtr, N, seed=17))
    for r in range(world):
        assert abi.relerr(np.load(tmp_path / f"ls{r}.npy"), want["ls"]) < 2e-6
        assert abi.relerr(np.load(tmp_path / f"ts{r}.npy"), want["tsPWS"]) < 2e-6


@pytest.mark.parametrize("world", [2, 3])
def test_scale_sharded_finish_over_gloo(tmp_path, world, monkeypatch):
    """stack_sharded(schedule="sharded-finish"), the scale-sharded finish stage: the ranks finish
    disjoint runs of octaves and add their partial reconstructions; with TSPWS_SHARD_FINISH=0 every rank finishes
    redundantly.  Both schedules against the unsharded oracle call."""
    kw, mtr, N = dict(Kmax=5, unbiased=1), 23, 1024
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), abi.synth_traces(mtr, N, seed=17))
    monkeypatch.setenv("TEST_SCHEDULE", "sharded-finish")
    for mode in ("1", "0"):
        monkeypatch.setenv("TSPWS_SHARD_FINISH", mode)
        out = tmp_path / mode
        out.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), kw, mtr, N, str(out)), nprocs=world, join=True)
        for r in range(world):
            assert abi.relerr(np.load(out / f"ls{r}.npy"), want["ls"]) < 2e-6
            assert abi.relerr(np.load(out / f"ts{r}.npy"), want["tsPWS"]) < 2e-6


def test_empty_shard_reaches_the_collective(tmp_path):
    """mtr_global < world: rank 0 holds no trace at all, yet it must walk the same sequence of collectives (the
    overlapped two-half reduction) and end with the same outputs as every other rank."""
    kw, mtr, N, world = dict(Kmax=2, unbiased=1), 2, 1024, 3
    assert tspws.shard_range(mtr, 0, world) == (0, 0)
    mp.spawn(_worker, args=(world, _free_port(), kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), abi.synth_traces(mtr, N, seed=17))
    for r in range(world):
        assert abi.relerr(np.load(tmp_path / f"ls{r}.npy"), want["ls"]) < 2e-6
        assert abi.relerr(np.load(tmp_path / f"ts{r}.npy"), want["tsPWS"]) < 2e-6


@pytest.mark.parametrize("schedule", ["single", "split", "sharded-finish"])
def test_three_schedules_over_gloo_cpu(tmp_path, schedule, monkeypatch):
    """The three placements of the one logical reduction (stack_sharded(schedule=...), bench.py --schedule, TSPWS_SCHEDULE): ONE
    all-reduce + redundant finish (north_star's wording), two halves overlapped with streaming / transforms, pieces + scale-sharded
    finish.  World 3 with ragged shards; every rank must end with the unsharded oracle call's outputs, and an unknown name is refused."""
    kw, mtr, N, world = dict(Kmax=5, unbiased=1), 23, 1024, 3
    monkeypatch.setenv("TSPWS_SCHEDULE", schedule)
    mp.spawn(_worker, args=(world, _free_port(), kw, mtr, N, str(tmp_path)), nprocs=world, join=True)
    want = abi.run_main(abi.oracle().orc_tspws_main, abi.default_params(**kw), abi.synth_traces(mtr, N, seed=17))
    for r in range(world):
        assert abi.relerr(np.load(tmp_path / f"ls{r}.npy"), want["ls"]) < 2e-6
        assert abi.relerr(np.load(tmp_path / f"ts{r}.npy"), want["tsPWS"]) < 2e-6
    with pytest.raises(tspws.TspwsError):
        tspws.stack_sharded(OraclePlan(abi.default_params(**kw), N), torch.zeros((2, N)), 0, 2, schedule="ring")
