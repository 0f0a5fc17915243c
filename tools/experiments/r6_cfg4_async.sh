#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ for i in 1 2 3; do CFG4_REPS=30 python tools/cfg4_run.py; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_cfg4_async.txt
timeout 2400 python -m pytest tests/test_spectral_gpu.py tests/test_hip_parity.py tests/test_comm_gpu.py tests/test_cfg5_gpu.py -q -x -k "jackknife or masked or subsampl or replica" 2>&1 | tail -4
