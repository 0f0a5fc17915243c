#!/bin/bash
# round 6: cfg4 -- the inverses of the FIR kernels' octaves behind those kernels, beside the spectral chain's tail (TSPWS_JK_EARLY_INV=0: all behind the chain)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
CFG4_REPS=30 python tools/cfg4_run.py
TSPWS_LIB_PATH=$S CFG4_REPS=30 python tools/cfg4_run.py
TSPWS_LIB_PATH=$S TSPWS_JK_EARLY_INV=0 CFG4_REPS=30 python tools/cfg4_run.py
CFG4_REPS=30 python tools/cfg4_run.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_early_inv.txt
timeout 1200 python -m pytest tests/test_spectral_gpu.py tests/test_hip_parity.py -q -x -k "jackknife or masked or subsampl or replica" 2>&1 | tail -4
bash tools/gpu_timeline_cfg.sh r6cfg4 30 tools/cfg4_run.py | grep -v amdgpu.ids | tail -32
