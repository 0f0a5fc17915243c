#!/usr/bin/env python3
"""bench.py -- stacked samples/s of the ts-PWS hot path on MI355X.

Workload (BASELINE.json `metric`: "Morlet ts-PWS, 10k x 131072", configs[2]): per GPU 10 000 synthetic traces x 131 072
samples, default Morlet frame (V=4, J=14, 56 scales), two-stage stack with 10 groups + unbiased phase coherence.  One step =
one whole tspws_main-equivalent call on HBM-resident traces: partial stacks -> (all-reduce when N>1) -> 10 forward frame
CWTs + phase stack -> weight -> 2 inverse CWTs -> float outputs.  Weak scaling: every rank holds its own shard of an
(N x traces)-trace ensemble.  `--config cfg5` is BASELINE configs[4]: 12 500 traces per GPU (100 000 at N=8).

    python bench.py                         one GPU: headline line + roofline + CPU baseline + the other configs
    python bench.py --gpus 8                starts its own 8 ranks (torch.distributed.run, one per GPU, RCCL) -- no GPU call
                                            is made in the parent; under an external torchrun (WORLD_SIZE set) it just runs
    python bench.py --gpus 8 --config cfg5  BASELINE configs[4]

Prints ONE JSON line (the driver contract).  N=1 adds: `cpu_baseline` (the reference OpenMP path on this host),
`end_to_end_host_path` (the drop-in tspws_main on host buffers, PCIe included) and `other_configs` (cfg2 single-stage,
cfg4 Mexican hat + jackknife, each with its roofline fraction and its error against the reference on a bounded sub-batch).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (same guide); ~62 TFLOP/s is what a dense v_fma_f64 stream sustains


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["cfg3", "cfg5"], default="cfg3",
                    help="cfg3: 10 000 traces per GPU (BASELINE configs[2], the metric's config); cfg5: 12 500 per GPU (configs[4]: 100k over 8)")
    ap.add_argument("--traces", type=int, default=None, help="traces per GPU (overrides --config)")
    ap.add_argument("--samples", type=int, default=131072)
    ap.add_argument("--kmax", type=int, default=10)
    ap.add_argument("--schedule", choices=["single", "split", "sharded-finish"], default=None,
                    help="N > 1: placement of the one logical reduction of P[K][N] (ts-pws_amd.stack_sharded): single = ONE all-reduce + redundant "
                         "finish (north_star's wording; default), split = two halves overlapped with streaming / transforms, sharded-finish = "
                         "pieces K-2 | 2 + scale-sharded finish; the other two are timed in the same run as well (`schedules`)")
    ap.add_argument("--shard-of", type=int, default=None,
                    help="one GPU only: treat the traces as rank 0's shard of an ensemble this many times larger (cfg5 on one GPU: 12 500 of 100 000; "
                         "default 8 for --config cfg5 --gpus 1, else 1)")
    ap.add_argument("--strong-total", type=int, default=None,
                    help="N > 1: after the weak-scaling legs, one more leg with this many traces IN ALL, sharded over the ranks (strong scaling; default "
                         "100 000 = BASELINE configs[4] when the per-GPU size is the default one, else off; 0: off)")
    ap.add_argument("--collective-timeout", type=int, default=180, help="N > 1: seconds after which a hung collective fails the run")
    ap.add_argument("--leg-deadline", type=int, default=60,
                    help="N > 1: seconds an A/B leg (another schedule, the strong-scaling leg) may take before rank 0 prints the line it has and every rank exits")
    ap.add_argument("--inject-hang", choices=["single", "split", "sharded-finish", "strong"], default=None,
                    help="TEST ONLY (tests/test_cli_gpu.py): the last rank never enters this optional leg -- its peers sit in the leg's first collective until "
                         "the deadline prints the line measured so far")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline, host-path and other-config legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the host-path and other-config legs only")
    return ap.parse_args()


def self_launch(args):
    """--gpus N without a launcher around us: start N ranks (one per GPU) BEFORE anything touches the GPU in this process,
    pass their rank-0 JSON line through and exit with their code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env["BENCH_SELF_LAUNCHED"] = "1"
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in out.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line)
    return out.returncode if out.returncode else (0 if line is not None else 1)


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if args.gpus > 1 and world_env != args.gpus:
        if os.environ.get("BENCH_SELF_LAUNCHED"):
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env} inside the launcher")
        if os.environ.get("BENCH_BACKEND", "nccl") == "nccl":
            import torch   # (counting devices does not initialise the GPU in this process)
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s): one rank per GPU over RCCL needs {args.gpus}")
        sys.exit(self_launch(args))
    run(args)


def stats(v):
    import numpy as np
    v = np.asarray(v, dtype=np.float64)
    return {"median": float(np.median(v)), "min": float(v.min()), "max": float(v.max()), "mean": float(v.mean()), "n": int(v.size)} if v.size else None


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import abi
    tspws = importlib.import_module("ts-pws_amd")

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    backend = os.environ.get("BENCH_BACKEND", "nccl")  # "gloo" lets the N>1 path be smoke-tested on a 1-GPU box
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        # (a schedule that hangs -- ranks issuing different collective sequences -- must fail the run, not eat the caller's lease)
        tmo = datetime.timedelta(seconds=max(30, args.collective_timeout))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    mtr_local = args.traces if args.traces is not None else (12500 if args.config == "cfg5" else 10000)
    N, K = args.samples, args.kmax
    # one GPU, cfg5: the traces are rank 0's shard of the 8-times larger ensemble of BASELINE configs[4] -- the group index comes
    # from the GLOBAL trace index (ts_pws1f_lib.c:876), so the shard fills groups 0 and 1 and leaves zero rows, as on a node
    shard_of = args.shard_of if args.shard_of is not None else (8 if (args.config == "cfg5" and world == 1) else 1)
    assert shard_of == 1 or world == 1, "--shard-of is the one-GPU stand-in for a shard of a larger ensemble"
    mtr_global = mtr_local * world * shard_of
    first = rank * mtr_local
    # default placement of the reduction: north_star's wording -- ONE all-reduce between the halves.  The other two placements are timed in
    # the same run (`schedules`), each under a deadline, so that the first run on real xGMI links A/Bs them without betting the headline
    # on a collective sequence that has only ever run over gloo / with one RCCL rank
    schedule = args.schedule or os.environ.get("TSPWS_SCHEDULE") or "single"
    params = tspws.resolve(abi.default_params(Kmax=K, unbiased=1), N)
    t_plan = time.perf_counter()
    plan = tspws.Plan(params, N, device=local)  # frame geometry + tap generation on the device (outside the timed region)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan
    X = tspws.synth(mtr_local, N, seed=1, first=first, device=local)
    lib = tspws.load()
    ls = torch.empty(N, dtype=torch.float32, device=X.device)
    ts = torch.empty(N, dtype=torch.float32, device=X.device)
    red = plan.reduce_buffer(mtr_global)

    single = world == 1 and shard_of == 1
    x2 = torch.empty(2 * N, dtype=torch.float64, device=X.device)

    def make_leg(sched, Xl, first_l, glob_l):
        """One timed configuration: a schedule on a shard.  Returns (step, events, launch counter, the schedule that actually runs)."""
        # N > 1: share of the scales this rank finishes (None: no sharded finish -- agreed across the ranks)
        shard = tspws._finish_shard(plan, glob_l) if (world > 1 and sched == "sharded-finish") else None
        if world > 1 and sched == "sharded-finish" and shard is None:
            sched = "split"   # this frame / these parameters have no sharded finish
        # events on the launch stream of every timed step: start, end of streaming, reductions done (finish may start), end
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        launches = [0]
        buf = plan.reduce_buffer(glob_l).view(K, N)

        def piece(g0, g1, count):
            plan.partial_stacks_range(Xl, first_l, glob_l, g0, g1)
            if count:
                launches[0] += lib.tspws_hip_stream_launches(plan.h)

        def step(i=None, count=False):
            if single:
                # one GPU: tspws_hip_stack = stack_local + stack_finish in one C call; HIP events inside the library bracket the
                # call and its streaming stage on the launch stream
                plan.stack_single(Xl, ls, ts)
                return
            if i is not None:
                ev[i][0].record()
            if world == 1:
                # one GPU standing in for one rank of a larger job: the shard-local half with the global group index, no collective
                # (there is no peer), then the finish stage on the shard's own buffer
                plan.stack_local(Xl, first_l, glob_l)
                if count:
                    launches[0] += lib.tspws_hip_stream_launches(plan.h)
                if i is not None:
                    ev[i][1].record()
                    ev[i][2].record()
                plan.stack_finish(glob_l, ls, ts)
            elif sched == "single" or K < 2:
                # north_star's wording: ONE all-reduce of the whole buffer between the halves, every rank finishes redundantly
                piece(0, K, count)
                if i is not None:
                    ev[i][1].record()
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                if i is not None:
                    ev[i][2].record()
                plan.stack_finish(glob_l, ls, ts)
            else:
                # the streaming stage in two pieces of the groups; the all-reduce of the first piece (RCCL, its own stream) overlaps the
                # streaming of the second -- still one logical fp64 reduction of P[Kmax][N] (ts-pws_amd.stack_sharded)
                half = max(1, tspws.split_groups(K, shard is not None))
                piece(0, half, count)
                w1 = dist.all_reduce(buf[:half], op=dist.ReduceOp.SUM, async_op=True)
                piece(half, K, count)
                if i is not None:
                    ev[i][1].record()
                w2 = dist.all_reduce(buf[half:], op=dist.ReduceOp.SUM, async_op=True)
                if shard is not None:
                    # scale-sharded finish: this rank transforms / weights / reconstructs its share of the scales only; the ranks
                    # add their partial reconstructions (2 N doubles) and every rank ends with the outputs
                    w1.wait()
                    w2.wait()
                    if i is not None:
                        ev[i][2].record()
                    plan.stack_finish_scales(glob_l, shard[0], shard[1], x2)
                    dist.all_reduce(x2, op=dist.ReduceOp.SUM)
                    plan.epilogue(x2, glob_l, ls, ts)
                else:
                    w1.wait()
                    plan.stack_finish_range(glob_l, 0, half)   # transforms of the reduced half run beside the second reduction
                    w2.wait()
                    if i is not None:
                        ev[i][2].record()
                    plan.stack_finish_range(glob_l, half, K)
                    plan.stack_finish_tail(glob_l, ls, ts)
            if i is not None:
                ev[i][3].record()
        return step, ev, launches, sched

    def time_leg(step, warm):
        """warm untimed steps, then EXACTLY args.steps timed ones between barrier + synchronize on both sides; MAX over the ranks"""
        for w in range(max(1, warm)):
            step(count=(w == 0))
        torch.cuda.synchronize()
        if single:
            plan.profile_begin(args.steps)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=X.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def rank_columns(ev):
        """where a step's time goes on every rank (HIP events on the launch stream): streaming, the part of the reduction(s) that is NOT
        hidden behind streaming or transforms (end of streaming -> all partial stacks reduced), the rest (finish stage incl. its own small
        all-reduce under sharded-finish)"""
        mine = [float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev])),
                float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))]
        if world > 1:
            t3 = torch.tensor(mine, dtype=torch.float64, device=X.device)
            allr = [torch.zeros_like(t3) for _ in range(world)]
            dist.all_gather(allr, t3)
            return [[round(float(v), 4) for v in t.tolist()] for t in allr]
        return [[round(v, 4) for v in mine]]

    def digest():
        import hashlib
        torch.cuda.synchronize()
        return hashlib.sha1(ls.cpu().numpy().tobytes() + ts.cpu().numpy().tobytes()).hexdigest()[:16]

    step, ev, launches, schedule = make_leg(schedule, X, first, mtr_global)
    dt = time_leg(step, args.warmup)
    out_sha = digest() if world > 1 else None

    if single:
        stage_ms, call_ms = plan.profile_read()
        nlaunch = max(1, lib.tspws_hip_stream_launches(plan.h))
    else:
        stage_ms = np.array([e[0].elapsed_time(e[1]) for e in ev])
        call_ms = np.array([e[0].elapsed_time(e[3]) for e in ev])
        nlaunch = max(1, launches[0])
    stream_ms = float(np.mean(stage_ms))
    alg_bytes = 4.0 * mtr_local * N + 8.0 * K * N   # read every float32 sample once + write the K fp64 partials
    achieved = alg_bytes / (stream_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes of profiles/collect_pmc.sh (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs): a record
    # is quoted only for the exact launch shape AND the exact source of the streaming kernel it was measured on (sha256 of
    # csrc/stream.hip, written by profiles/collect_round.sh) -- a changed kernel reports null until its counters are collected again
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_partial_stacks.json")
    if os.path.exists(tf) and (mtr_local, N, K, shard_of, world) == (10000, 131072, 10, 1, 1):   # (N > 1: null -- the counters were collected on one GPU)
        try:
            rec = json.load(open(tf))
            same_src = rec.get("stream_hip_sha256") == file_sha256(os.path.join(ROOT, "ts-pws_amd", "csrc", "stream.hip"))
            traffic = rec.get("hbm_bytes_per_launch") if (rec.get("launches_per_call") == nlaunch and same_src) else None
        except Exception:
            traffic = None

    which = ("BASELINE configs[4]-shard: one rank's 12 500 of the 100 000 traces (global group index, rows of foreign groups zero), no peer to reduce with"
             if (mtr_local == 12500 and world == 1 and shard_of == 8) else
             "BASELINE configs[4]: 100k traces over 8 GPUs" if (mtr_local == 12500 and world == 8) else
             "BASELINE configs[4] shard size, other rank count" if mtr_local == 12500 else
             "BASELINE configs[2]" if (mtr_local, N, K) == (10000, 131072, 10) else "custom size")
    if world == 1:
        par = "one GPU: no collective" + (f" (rank 0 of a virtual {shard_of}-rank job: shard-local half + finish)" if shard_of > 1 else "")
    elif schedule == "sharded-finish":
        par = (f"schedule=sharded-finish, trace-sharded x{world}: fp64 all-reduce of P[K][N] in two pieces (the first, K-2 groups, overlaps the streaming of the last two), then a "
               f"scale-sharded finish stage: every rank transforms / weights / reconstructs its share of the scales, all-reduce of the 2 N partial reconstructions")
    elif schedule == "split":
        par = (f"schedule=split, trace-sharded x{world}: fp64 all-reduce of P[K][N] in two halves: the first overlaps the streaming of the second, the second the "
               f"transforms of the first; every rank finishes redundantly")
    else:
        par = f"schedule=single, trace-sharded x{world}: ONE fp64 all-reduce of P[K][N] between the shard-local half and the redundant finish stage"
    res = {
        "metric": baseline_metric(),
        "value": mtr_local * world * N * args.steps / dt,   # the samples all ranks processed (a one-GPU shard of a larger ensemble counts its own traces only)
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{mtr_local} traces/GPU x {N} samples, Morlet w0=pi*sqrt(2/ln2) V=4 J={params.J}, "
                               f"two-stage K={K} + unbiased phase coherence ({which}); HBM-resident float32 traces",
                   "traces_total": mtr_global, "traces_per_gpu": mtr_local, "plan_create_ms": round(t_plan * 1e3, 3), "parallelism": par,
                   "world_size": dist.get_world_size() if world > 1 else 1, "backend": (backend if world > 1 else None),
                   "rccl_version": rccl_version(torch) if (world > 1 and backend == "nccl") else None},
        "step_ms_gpu": stats(call_ms),   # per call, HIP events on the launch stream (rank 0): median / min / max over the timed steps
        "roofline": {"bound": "hbm", "kernel": "k_partial", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes / nlaunch, "ms_per_launch": stream_ms / nlaunch, "launches_per_call": nlaunch,
                     "stage_ms": stats(stage_ms),
                     "note": "HIP events on the launch stream around the streaming stage of every timed call: back-to-back k_partial "
                             "launches (one workgroup per CU), every group written directly; per-launch = stage / launches"},
        "whole_call_frac_of_hbm_roofline": (alg_bytes / (dt / args.steps) / 1e9) / HBM_PEAK_GBS,
    }

    if not single:
        res["per_rank_ms"] = {"columns": ["stream", "exposed_collective", "finish"], "ranks": rank_columns(ev), "schedule": schedule if world > 1 else None}
    if world > 1 and rank == 0 and not args.no_cpu:
        # N > 1: the reference OpenMP path on THIS node's host cores beside the N-GPU number, in the same run (north_star): rank 0
        # runs it on its own shard after the timed loop (the other ranks wait in the barrier below), and checks its shard-local
        # partial stacks -- global group index, before any reduction -- against direct FP64 sums of the same traces
        res["cpu_baseline"] = cpu_baseline_shard(abi, plan, X, abi.default_params(Kmax=K, unbiased=1), N, K, mtr_local, first, mtr_global, red)
    if world > 1:
        # The driver's ONE multi-GPU run as an A/B: every schedule in the same invocation -- same plan, same traces, args.steps timed
        # steps each between barriers.  `value` / `ms_per_step` above stay the default schedule's; `single` (north_star's wording) and
        # `split` end in the one-GPU call's accumulation order, so their outputs are bit-identical to it (output_sha1: ls || tsPWS on rank 0).
        legs = {}
        res["schedules"] = legs

        class Deadline:
            """An optional leg must not cost the run its headline: when it overruns (a hung collective blocks in synchronize, which releases
            the GIL), rank 0 prints the line measured so far with the leg named, and every rank leaves with status 0."""
            def __init__(self, label):
                import threading
                self.label = label
                self.t = threading.Timer(max(5, args.leg_deadline), self.bail)
                self.t.daemon = True
            def bail(self):
                if rank == 0:
                    res["aborted_leg"] = {"leg": self.label, "deadline_s": args.leg_deadline}
                    sys.stdout.write(json.dumps(res) + "\n")
                    sys.stdout.flush()
                os._exit(0)
            def __enter__(self):
                self.t.start()
                return self
            def __exit__(self, et, ev_, tb):
                self.t.cancel()
                if et is not None and issubclass(et, Exception):
                    # a leg that FAILS (not hangs) on this rank: the peers sit in its next collective until their own deadline; this rank
                    # leaves the same way -- rank 0 with the line it has
                    if rank == 0:
                        res["aborted_leg"] = {"leg": self.label, "error": repr(ev_)[:500]}
                        sys.stdout.write(json.dumps(res) + "\n")
                        sys.stdout.flush()
                    os._exit(0)
                return False

        for sc in ("single", "split", "sharded-finish"):
            if sc == schedule:
                legs[sc] = {"ms_per_step": dt / args.steps * 1e3, "value": mtr_local * world * N * args.steps / dt, "per_rank_ms": res["per_rank_ms"]["ranks"],
                            "runs_as": schedule, "output_sha1": out_sha, "is_default": True}
                continue
            with Deadline(f"schedule {sc}"):
                if args.inject_hang == sc and rank == world - 1:
                    time.sleep(10 ** 6)
                st2, ev2, _, eff = make_leg(sc, X, first, mtr_global)
                d2 = time_leg(st2, min(2, max(1, args.warmup)))
                cols2, sha2 = rank_columns(ev2), digest()
            legs[sc] = {"ms_per_step": d2 / args.steps * 1e3, "value": mtr_local * world * N * args.steps / d2, "per_rank_ms": cols2,
                        "runs_as": eff, "output_sha1": sha2, "is_default": False}
        # Strong scaling beside the weak figures: a FIXED ensemble (BASELINE configs[4]: 100 000 traces in all) sharded over the ranks
        total = args.strong_total if args.strong_total is not None else (100000 if (args.traces is None and args.config == "cfg3") else 0)
        if total and total >= world:
            lo, cnt = tspws.shard_range(total, rank, world)
            with Deadline("strong scaling"):
                if args.inject_hang == "strong" and rank == world - 1:
                    time.sleep(10 ** 6)
                X5 = tspws.synth(cnt, N, seed=1, first=lo, device=local)
                st5, ev5, _, eff5 = make_leg(schedule, X5, lo, total)
                d5 = time_leg(st5, min(2, max(1, args.warmup)))
                cols5 = rank_columns(ev5)
            res["strong_scaling"] = {"workload": f"{total} traces IN ALL x {N} samples over {world} GPUs (BASELINE configs[4] when 100 000 x 131 072), same frame and "
                                                 f"parameters; rank r holds the contiguous shard tspws_shard_range(total, r, world)",
                                     "traces_total": total, "traces_on_rank0": cnt if rank == 0 else None, "schedule": eff5, "ms_per_step": d5 / args.steps * 1e3,
                                     "value": total * N * args.steps / d5, "unit": "samples/s", "scaling": "strong", "per_rank_ms": cols5}
            del X5
    if world == 1 and shard_of == 1 and rank == 0:
        res["with_output_d2h"] = with_d2h(torch, plan, X, ls, ts, N, mtr_local, args.steps)
        if not args.no_cpu:
            Xh = X.cpu().numpy()
            res["cpu_baseline"], ref_out = cpu_baseline(abi, Xh, abi.default_params(Kmax=K, unbiased=1), ls, ts, N, mtr_local)
            if not args.no_extra:
                res["end_to_end_host_path"] = host_path(abi, lib, Xh, abi.default_params(Kmax=K, unbiased=1), N, mtr_local, ref_out,
                                                        res["cpu_baseline"].get("seconds"))
                del Xh
                res["other_configs"] = other_configs(abi, tspws, lib, torch, X, N)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def file_sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def rccl_version(torch):
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def baseline_metric():
    """The metric string of BASELINE.json (its `metric` is quoted on the 10k x 131072 Morlet config this bench runs)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "stacked samples/s (Morlet ts-PWS, 10k×131072) + % HBM roofline, 1/2/4/8 GPU"


def with_d2h(torch, plan, X, ls, ts, N, mtr, steps):
    """SURVEY 8d's protocol also moves the two float outputs to the host inside the step: the same loop with both copies
    (pinned buffers, same stream) after every call.  A labelled second figure -- `value` stays the HBM-resident rate."""
    hl = torch.empty(N, dtype=torch.float32).pin_memory()
    ht = torch.empty(N, dtype=torch.float32).pin_memory()
    for _ in range(2):
        plan.stack_single(X, ls, ts)
        hl.copy_(ls, non_blocking=True)
        ht.copy_(ts, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.stack_single(X, ls, ts)
        hl.copy_(ls, non_blocking=True)
        ht.copy_(ts, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_step": dt * 1e3, "value": mtr * N / dt, "unit": "samples/s", "note": "each step ends with the D2H copies of ls and tsPWS (2 x N floats)"}


def call_main(abi, fn, params_in, Xh, N, mtr, times=None, C_rep=0):
    """One tspws_main-shaped call on host traces WITHOUT copying them (the call may rewrite them: fold / rm are off here)."""
    import numpy as np
    p = abi.t_tsPWS.from_buffer_copy(params_in)
    out = abi.t_tsPWS_out()
    l = np.zeros(N, np.float32)
    t = np.zeros(N, np.float32)
    fp = C.POINTER(C.c_float)
    out.ls, out.tsPWS = l.ctypes.data_as(fp), t.ctypes.data_as(fp)
    out.N, out.mtr = N, mtr
    keep = []
    jl = jt = jm = None
    if C_rep:
        jl, jt, jm = np.zeros((C_rep, N), np.float32), np.zeros((C_rep, N), np.float32), np.zeros(C_rep, np.uint32)
        rows_l = (fp * C_rep)(*[jl[c].ctypes.data_as(fp) for c in range(C_rep)])
        rows_t = (fp * C_rep)(*[jt[c].ctypes.data_as(fp) for c in range(C_rep)])
        keep += [rows_l, rows_t]
        out.ls_subsmpl, out.tsPWS_subsmpl = C.cast(rows_l, C.POINTER(fp)), C.cast(rows_t, C.POINTER(fp))
        out.mtr_subsmpl = jm.ctypes.data_as(C.POINTER(C.c_uint))
        out.M = C_rep
    d = abi.t_data()
    d.sigall = Xh.ctypes.data_as(fp)
    if times is not None:
        d.time = times.ctypes.data_as(C.POINTER(abi.time_t))
    d.hdr.max, d.hdr.mtr, d.hdr.dt, d.hdr.beg = N, mtr, 1.0, 0.0
    t0 = time.perf_counter()
    rc = fn(C.byref(p), C.byref(out), C.byref(d))
    sec = time.perf_counter() - t0
    return {"rc": rc, "seconds": sec, "ls": l, "tsPWS": t, "jk_ls": jl, "jk_ts": jt, "jk_mtr": jm}


def cpu_baseline(abi, Xh, params_in, ls, ts, N, mtr):
    """Time the CPU path on this host on the SAME traces: the reference itself when oracle/_ref was built (kind "reference"),
    else this repo's restatement (kind "port").  Bounded sample: at most 10 000 traces (~1-3 s for the reference)."""
    import numpy as np
    ref = abi.ref()
    fn, kind = (ref.tspws_main, "reference") if ref is not None else (abi.oracle().orc_tspws_main, "port")
    n = min(mtr, 10000)
    r = call_main(abi, fn, params_in, Xh[:n], N, n)
    base = {"value": n * N / r["seconds"], "unit": "samples/s", "cores": os.cpu_count(), "kind": kind, "seconds": r["seconds"], "rc": r["rc"],
            "sample": f"{n} x {N} of the same synthetic traces (copied from HBM), whole tspws_main call, OpenMP team = all cores "
                      f"(the reference's trace loop is serial, so ~1 core does the work)"}
    if n == mtr:  # full-size parity of the GPU result against the CPU result on identical inputs
        base["gpu_vs_cpu_relerr"] = {"ls": abi.relerr(ls.cpu().numpy(), r["ls"]), "tsPWS": abi.relerr(ts.cpu().numpy(), r["tsPWS"]),
                                     "max_abs_ls": float(np.abs(r["ls"]).max()), "max_abs_tsPWS": float(np.abs(r["tsPWS"]).max())}
    # the honest stronger host baseline (line ii): this repo's restatement with the trace loop,
    # the per-scale transforms and the inverse spread over all cores (bit-identical to the serial restatement)
    r2 = call_main(abi, abi.oracle().orc_tspws_main_mt, params_in, Xh[:n], N, n)
    base["parallel_port"] = {"value": n * N / r2["seconds"], "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "seconds": r2["seconds"], "rc": r2["rc"],
                             "sample": "same traces; trace-/scale-parallel OpenMP restatement (oracle/tspws_oracle.c: orc_tspws_main_mt)",
                             "relerr_vs_reference": {"ls": abi.relerr(r2["ls"], r["ls"]), "tsPWS": abi.relerr(r2["tsPWS"], r["tsPWS"])}}
    return base, (r if n == mtr else None)


def cpu_baseline_shard(abi, plan, X, params_in, N, K, mtr_local, first, mtr_global, red):
    """world > 1, rank 0: the CPU path (reference when oracle/_ref is built, else the restatement) on this rank's own shard -- a
    bounded sample of the job: 1 / world of the traces --, and the shard-vs-CPU error of the rank's partial-stack buffer."""
    import numpy as np
    import torch
    ref = abi.ref()
    fn, kind = (ref.tspws_main, "reference") if ref is not None else (abi.oracle().orc_tspws_main, "port")
    Xh = X.cpu().numpy()
    r = call_main(abi, fn, params_in, Xh, N, mtr_local)
    base = {"value": mtr_local * N / r["seconds"], "unit": "samples/s", "cores": os.cpu_count(), "kind": kind, "seconds": r["seconds"], "rc": r["rc"],
            "sample": f"rank 0's shard: {mtr_local} x {N} of the job's {mtr_global} synthetic traces (copied from HBM), whole tspws_main call on the shard, "
                      f"OpenMP team = all cores (the reference's trace loop is serial, so ~1 core does the work)"}
    # shard-local partial stacks (ts_pws1f_lib.c:866-881 with the GLOBAL trace index) against direct FP64 sums on the host
    plan.stack_local(X, first, mtr_global)
    torch.cuda.synchronize()
    got = red.view(K, N).cpu().numpy()
    g = ((first + np.arange(mtr_local, dtype=np.int64)) * K) // mtr_global
    want = np.zeros((K, N), np.float64)
    for k in np.unique(g):
        idx = np.nonzero(g == k)[0]
        want[k] = np.add.reduce(Xh[idx[0]:idx[-1] + 1], axis=0, dtype=np.float64)
    base["shard_partial_stacks_relerr"] = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
    return base


def host_path(abi, lib, Xh, params_in, N, mtr, ref_out, ref_seconds):
    """The drop-in itself: tspws_main on HOST buffers -- frame creation, pinning + upload over PCIe, compute, download.  Two
    calls: the first builds the frame, the second finds it (and the device trace buffer) in tspws_main's cache."""
    a = call_main(abi, lib.tspws_main, params_in, Xh, N, mtr)
    b = call_main(abi, lib.tspws_main, params_in, Xh, N, mtr)
    lib.tspws_main_release()
    res = {"first_call_s": a["seconds"], "cached_call_s": b["seconds"], "rc": [a["rc"], b["rc"]],
           "value": mtr * N / b["seconds"], "unit": "samples/s", "input_GBps": 4.0 * mtr * N / b["seconds"] / 1e9,
           "note": "whole tspws_main call on host memory (PCIe-bound: 4 B per sample must cross the link); never the headline value"}
    if ref_out is not None:
        res["relerr_vs_reference"] = {"ls": abi.relerr(b["ls"], ref_out["ls"]), "tsPWS": abi.relerr(b["tsPWS"], ref_out["tsPWS"])}
        if ref_seconds:
            res["speedup_vs_reference_host_path"] = ref_seconds / b["seconds"]
    return res


def timeit(torch, fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def other_configs(abi, tspws, lib, torch, X, N):
    """The other single-GPU configs of BASELINE.json, after the headline timing: cfg2 (configs[1], single-stage, FP64-bound) and
    cfg4 (configs[3], Mexican hat + jackknife), each with its roofline fraction and its error against the reference (or the
    restatement where oracle/_ref is absent) on a bounded sub-batch of the same traces."""
    import numpy as np
    ref = abi.ref()
    cpu_fn, kind = (ref.tspws_main, "reference") if ref is not None else (abi.oracle().orc_tspws_main, "port")
    out = {}
    # ---- cfg2: 1024 x 32768, w0 = 2 pi, single-stage, wu = 2 biased -----------------------------------------------------
    N2, m2 = 32768, 1024
    pin = abi.default_params(w0=2 * np.pi)
    p2 = tspws.resolve(pin, N2)
    pl2 = tspws.Plan(p2, N2)
    X2 = tspws.synth(m2, N2, seed=1)
    l2 = torch.empty(N2, dtype=torch.float32, device="cuda")
    t2 = torch.empty(N2, dtype=torch.float32, device="cuda")
    # (the per-call time keeps falling over the first ~20 calls -- clocks ramp up under the FP64 load: 3 / 10 / 40 timed calls
    # after 2 warm-ups give 3.5 / 3.4 / 3.25 ms --, so the steady state is what is timed)
    sec = timeit(torch, lambda: pl2.stack_single(X2, l2, t2), 40, 10)
    tab = pl2.tables()
    macs = float(np.sum(tab["L"].astype(np.float64) * tab["Ns"].astype(np.float64)))  # complex-real MACs per transformed trace
    flops = 4.0 * macs * m2                      # what the FIR form of the whole frame costs
    # engine of this batch: the scales [sf, S) go through the traces' spectra (csrc/spectral.hip), the finer ones stay FIR sums
    sf = int(lib.tspws_hip_spectral_choice(pl2.h, m2))
    Ls, Nss = tab["L"].astype(np.float64), tab["Ns"].astype(np.float64)
    fir_part = 4.0 * float(np.sum(Ls[:sf] * Nss[:sf]))
    nspec = pl2.S - sf
    M2 = N2 // 2
    spec_part = 0.0
    if nspec:
        spec_part = (5.0 * M2 * np.log2(M2) + 10.0 * M2                  # one packed real transform per trace (N/2-point complex + split)
                     + 8.0 * N2 * nspec                                    # multiply-and-fold: 4 FMAs per (frequency, scale)
                     + float(np.sum(5.0 * Nss[sf:] * np.log2(np.maximum(Nss[sf:], 2.0)))))   # N_s-point inverse transforms
    executed = (fir_part + spec_part) * m2
    nsub = 128
    g = pl2.stack_single(X2[:nsub])
    torch.cuda.synchronize()
    r = call_main(abi, cpu_fn, pin, X2[:nsub].cpu().numpy(), N2, nsub)
    out["cfg2_single_stage_1024x32768_w2pi"] = {
        "ms_per_call": sec * 1e3, "timed_calls": 40, "warmup_calls": 10, "value": m2 * N2 / sec, "unit": "samples/s", "V": p2.V, "J": p2.J, "scales": pl2.S,
        "engine": ("fir" if not nspec else f"fir (scales 0..{sf - 1}: D <= {int(tab['D'][sf - 1])}) + spectral (scales {sf}..{pl2.S - 1}: D >= {int(tab['D'][sf])})"),
        "roofline": {"bound": "fp64 vector", "flops_per_call": executed, "achieved": executed / sec / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": executed / sec / 1e12 / FP64_PEAK_TFLOPS, "hbm_frac": 4.0 * m2 * N2 / sec / 1e9 / HBM_PEAK_GBS,
                     "flops_executed": {"fir_scales": fir_part * m2, "spectral_scales": spec_part * m2},
                     "fir_equivalent_flops_per_call": flops, "fir_equivalent_tflops": flops / sec / 1e12,
                     "note": "EXECUTED forward flops: FIR sums of the fine scales (4 flop per complex-real MAC) + transforms and multiply-and-fold of the "
                             "spectral scales; `fir_equivalent_*` is what the FIR form of the whole frame would cost (SURVEY 8d) over the same time -- a "
                             "speed figure comparable with earlier rounds, not a roofline fraction"},
        "check": {"kind": kind, "sample": f"first {nsub} traces, whole call", "cpu_seconds": r["seconds"],
                  "relerr": {"ls": abi.relerr(g[0].cpu().numpy(), r["ls"]), "tsPWS": abi.relerr(g[1].cpu().numpy(), r["tsPWS"])}}}
    del X2, pl2
    # ---- cfg1: the shipped example's shape, 499 x 16501 (N odd: no decimation divides it), single-stage and TwoStage=10 unbiased ---
    N1, m1 = 16501, 499
    X1 = tspws.synth(m1, N1, seed=1)
    X1h = X1.cpu().numpy()
    l1 = torch.empty(N1, dtype=torch.float32, device="cuda")
    t1 = torch.empty(N1, dtype=torch.float32, device="cuda")
    c1 = {}
    for name, kw in (("single_stage", dict()), ("two_stage_k10_unbiased", dict(Kmax=10, unbiased=1))):
        pin = abi.default_params(**kw)
        pl1 = tspws.Plan(tspws.resolve(pin, N1), N1)
        sec = timeit(torch, lambda: pl1.stack_single(X1, l1, t1), 40, 10)
        torch.cuda.synchronize()
        r = call_main(abi, cpu_fn, pin, X1h, N1, m1)   # the WHOLE ensemble on the CPU: this is the reference's own runnable size
        c1[name] = {"ms_per_call": sec * 1e3, "value": m1 * N1 / sec, "unit": "samples/s", "cpu_seconds": r["seconds"], "speedup_hbm_resident": r["seconds"] / sec,
                    "relerr": {"ls": abi.relerr(l1.cpu().numpy(), r["ls"]), "tsPWS": abi.relerr(t1.cpu().numpy(), r["tsPWS"])}}
        if name == "single_stage":
            # FP64 roofline of the single-stage call (the reference's default mode on its only data set): EXECUTED forward flops by engine --
            # FIR sums of the fine octaves (trace-lane kernel), the spectral chain of the far-decimated ones over a window of NT >= N + L - 1
            # samples of the periodic extension (N is odd: no decimation divides it), the clipped scales as a dense contraction (4 N N_s flop per
            # scale and trace on the matrix pipe) -- and what the FIR form of the whole frame costs, as for cfg2
            tab1 = pl1.tables()
            L1, Ns1 = tab1["L"].astype(np.float64), tab1["Ns"].astype(np.float64)
            sf1, se1 = int(lib.tspws_hip_spectral_choice(pl1.h, m1)), int(lib.tspws_hip_spectral_end_scale(pl1.h))
            NT1 = int(lib.tspws_hip_spectral_transform_length(pl1.h))
            fir_all = 4.0 * float(np.sum(L1 * Ns1))
            if sf1 < pl1.S:
                Mh = NT1 // 2
                Nb = NT1 / tab1["D"][sf1:se1].astype(np.float64)
                parts = {"fir_scales": 4.0 * float(np.sum(L1[:sf1] * Ns1[:sf1])),
                         "spectral_scales": 5.0 * Mh * np.log2(Mh) + 10.0 * Mh + 8.0 * NT1 * (se1 - sf1) + float(np.sum(5.0 * Nb * np.log2(np.maximum(Nb, 2.0)))),
                         "contraction_scales": 4.0 * N1 * float(np.sum(Ns1[se1:]))}
                eng = (f"fir (scales 0..{sf1 - 1}: D <= {int(tab1['D'][sf1 - 1])}) + spectral (scales {sf1}..{se1 - 1}: D >= {int(tab1['D'][sf1])}, transform window {NT1})"
                       + (f" + matrix-pipe contraction (scales {se1}..{pl1.S - 1}: filters longer than the window)" if se1 < pl1.S else ""))
            else:
                parts, eng = {"fir_scales": fir_all}, "fir"
            ex1 = sum(parts.values()) * m1
            c1[name]["engine"] = eng
            c1[name]["roofline"] = {"bound": "fp64 vector", "flops_per_call": ex1, "achieved": ex1 / sec / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": ex1 / sec / 1e12 / FP64_PEAK_TFLOPS, "hbm_frac": 4.0 * m1 * N1 / sec / 1e9 / HBM_PEAK_GBS,
                                    "flops_executed": {k: v * m1 for k, v in parts.items()},
                                    "fir_equivalent_flops_per_call": fir_all * m1, "fir_equivalent_tflops": fir_all * m1 / sec / 1e12,
                                    "note": "EXECUTED forward flops by engine (the spectral chain counts its padded transforms); `fir_equivalent_*` = the FIR form of "
                                            "the whole frame over the same time: a speed figure comparable with earlier rounds (round 5: 23.0 TFLOP/s), not a fraction"}
        del pl1
    out["cfg1_example_shape_499x16501"] = {"timed_calls": 40, "warmup_calls": 10, "check_kind": kind, "note": "BASELINE configs[0]: synthetic traces of the shipped "
                                           "example's shape; whole ensemble checked against the CPU path", **c1}
    del X1
    # ---- cfg4: 10k x 131072, Mexican hat, two-stage K = 10 + jackknife n = 10, d = 1 ---------------------------------------
    mtr = X.shape[0]
    pin = abi.default_params(type=-3, Kmax=10, jackknife_n=10, jackknife_d=1)
    p4 = tspws.resolve(pin, N)
    pl4 = tspws.Plan(p4, N)
    Cn = 10
    times = (1262304000 + 86400 * np.arange(mtr)).astype(np.int64)   # 2010-01-01 + i days
    sel = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel.ctypes.data, times.ctypes.data, mtr, 1, 10, Cn) == 0
    sec = timeit(torch, lambda: pl4.stack_jackknife(X, sel), 10, 3)
    # the class structure of a selection is kept per host thread (keyed by content): the figure above is the steady state of a
    # caller that repeats its time stamps (the same days for every station pair); a NEW selection pays the host-side class
    # construction again -- timed by alternating between two different selections
    times_b = times + 86400 * 5
    sel_b = np.zeros((Cn, mtr), np.int8)
    assert lib.tspws_jackknife_plan(sel_b.ctypes.data, times_b.ctypes.data, mtr, 1, 10, Cn) == 0
    flip = [0]

    def changed():
        flip[0] ^= 1
        pl4.stack_jackknife(X, sel_b if flip[0] else sel)
    sec_changed = timeit(torch, changed, 10, 2)
    alg = 4.0 * mtr * N + 8.0 * N + 8.0 * 10 * N + 8.0 * Cn * N
    tab4 = pl4.tables()
    macs4 = float(np.sum(tab4["L"].astype(np.float64) * tab4["Ns"].astype(np.float64)))
    flops4 = 4.0 * macs4 * 10 * (Cn + 1)   # K = 10 transforms for the stack and for every replica
    nsub = min(mtr, 1000)
    sel_s = np.zeros((Cn, nsub), np.int8)
    assert lib.tspws_jackknife_plan(sel_s.ctypes.data, times.ctypes.data, nsub, 1, 10, Cn) == 0
    g = pl4.stack_jackknife(X[:nsub], sel_s)
    torch.cuda.synchronize()
    r = call_main(abi, cpu_fn, pin, X[:nsub].cpu().numpy(), N, nsub, times=times[:nsub].copy(), C_rep=Cn)
    traffic4 = None   # whole-call counter traffic (profiles/pmc_record_call.py), quoted only for this size and the sources it was measured on
    tf4 = os.path.join(ROOT, "profiles", "pmc_cfg4_call.json")
    if os.path.exists(tf4) and (mtr, N) == (10000, 131072):
        try:
            import hashlib
            rec = json.load(open(tf4))
            hh = hashlib.sha256()
            for sname in rec.get("sources", []):
                hh.update(open(os.path.join(ROOT, "ts-pws_amd", "csrc", sname), "rb").read())
            traffic4 = rec.get("hbm_bytes_per_call") if hh.hexdigest() == rec.get("sources_sha256") else None
        except Exception:
            traffic4 = None
    out["cfg4_mexhat_twostage_jackknife_n10_d1"] = {
        "ms_per_call": sec * 1e3, "ms_per_call_changed_selection": sec_changed * 1e3,
        "selection": "ms_per_call: the same selection every call (class structure found in the per-thread memo); ms_per_call_changed_selection: a different selection every call",
        "timed_calls": 10, "warmup_calls": 3, "value": mtr * N / sec, "unit": "samples/s", "replicas": Cn, "traces": mtr, "V": p4.V, "J": p4.J, "scales": pl4.S,
        "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / sec / 1e9 / HBM_PEAK_GBS, "traffic": traffic4,
                     "note": "every sample read once (the stack and all replicas share ONE pass) + K partials + outputs; the 110 transforms "
                             "(3.6e10 flop) are as long as the stream"},
        "roofline_fp64": {"bound": "fp64 vector", "flops_per_call": flops4, "achieved": flops4 / sec / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": flops4 / sec / 1e12 / FP64_PEAK_TFLOPS,
                          "note": "the call's second bound: forward MACs of its 110 transforms (4 flop each) over the WHOLE call time -- the streaming walk "
                                  "and the transforms run one after the other, so neither fraction can approach 1; per-kernel figures: profiles/r05_timeline_cfg4.txt"},
        "check": {"kind": kind, "sample": f"first {nsub} traces, stack + {Cn} replicas", "cpu_seconds": r["seconds"],
                  "relerr": {"ls": abi.relerr(g[0].cpu().numpy(), r["ls"]), "tsPWS": abi.relerr(g[1].cpu().numpy(), r["tsPWS"]),
                             "jk_ls": abi.relerr(g[2].cpu().numpy(), r["jk_ls"]), "jk_ts": abi.relerr(g[3].cpu().numpy(), r["jk_ts"]),
                             "jk_mtr_equal": bool(np.array_equal(g[4], r["jk_mtr"]))}}}
    return out


if __name__ == "__main__":
    main()
