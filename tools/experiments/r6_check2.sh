#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_spectral_gpu.py -q -x > gpurun_out/r6_cli_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_cli_tests.log
tail -12 gpurun_out/r6_cli_tests.log
python tools/cfg_bench.py cfg1 40 2>&1 | grep -v amdgpu.ids
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py | grep -v amdgpu.ids | tail -26
