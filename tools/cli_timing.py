#!/usr/bin/env python3
"""Wall time of the command line on an ensemble of the shipped example's shape (499 SAC files x 16501 samples), phase by phase
(TSPWS_CLI_TIMES=1 prints them), next to the reference's library call on the same traces when oracle/_ref is present.
usage: cli_timing.py [files] [samples] [cli args ...]
Then the same ensemble 1, 2, 4 and 8 times in ONE process (`ts_pws @batch.txt`): wall time per additional list."""
import os, subprocess, sys, tempfile, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 499
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16501
extra = sys.argv[3:]
rng = np.random.default_rng(3)
with tempfile.TemporaryDirectory() as td:
    with open(os.path.join(td, "list.txt"), "w") as f:
        for i in range(n):
            abi.write_sac(os.path.join(td, f"t{i:04d}.sac"), rng.uniform(-0.5, 0.5, N).astype(np.float32), 1.0, 0.0, year=2010, jday=1 + i % 365)
            f.write(f"t{i:04d}.sac\n")
    cli = os.path.join(R, "ts-pws_amd", "bin", "ts_pws")
    for rep in range(3):
        t0 = time.perf_counter()
        out = subprocess.run([cli, "list.txt"] + extra, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, TSPWS_CLI_TIMES="1"))
        dt = time.perf_counter() - t0
        print(f"ts_pws {n} x {N} {' '.join(extra)}: {dt*1e3:.0f} ms wall, rc {out.returncode}; " + " ".join(l for l in out.stdout.splitlines() if l.startswith("cli:")))
    walls = {}
    for k in (1, 2, 4, 8):
        with open(os.path.join(td, "batch.txt"), "w") as f:
            for j in range(k):
                f.write(f"list.txt run{j}\n")
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            out = subprocess.run([cli, "@batch.txt"] + extra, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            best = min(best, time.perf_counter() - t0)
            assert out.returncode == 0, out.stdout
        walls[k] = best
        print(f"ts_pws @batch of {k} x ({n} x {N}) {' '.join(extra)}: {best*1e3:.0f} ms wall (best of 3)")
    print(f"per additional list: {(walls[8] - walls[1]) / 7 * 1e3:.1f} ms  (one-list process: {walls[1]*1e3:.0f} ms)")
