"""bench.py on a box WITHOUT a GPU: `--gpus N` must fail loudly -- non-zero exit, no JSON line, no hang -- because there is no CPU
stand-in for the path: over RCCL the parent counts the devices before it starts any rank and says so in one line; with the gloo
test backend (ranks may share a GPU) it starts its own ranks (no `WORLD_SIZE` assertion in the parent) and THEY fail."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present: tests/test_cli_gpu.py runs bench.py for real")
def test_bench_counts_the_devices_before_it_starts_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_SELF_LAUNCHED", "BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "--gpus 2 but this node has 0 GPU(s)" in out.stderr and "Traceback" not in out.stderr


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present: tests/test_cli_gpu.py runs bench.py for real")
def test_bench_starts_ranks_and_fails_loudly_without_a_device():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_SELF_LAUNCHED")}
    env["BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "WORLD_SIZE=1" not in out.stderr            # the parent did not reach the rank code: it launched children
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr or "elastic" in out.stderr
