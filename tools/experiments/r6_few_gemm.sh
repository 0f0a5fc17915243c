#!/bin/bash
# round 6, review item 3: the coarse scales of the ten partial stacks as a dense contraction (k_fwd_gemm_rows) instead of the direct kernel beside k_fwd_lds
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for rep in 1 2; do
python tools/finish_ab.py 60
TSPWS_FEW_GEMM=1 python tools/finish_ab.py 60
TSPWS_FEW_GEMM=1 TSPWS_FEW_GEMM_KS=64 python tools/finish_ab.py 60
TSPWS_FEW_GEMM=1 TSPWS_FEW_GEMM_KS=256 python tools/finish_ab.py 60
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_few_gemm.txt
TSPWS_FEW_GEMM=1 timeout 1200 python -m pytest tests/test_hip_parity.py -q -x -k "vs_oracle or golden or example" 2>&1 | tail -3
TSPWS_FEW_GEMM=1 bash tools/gpu_timeline_cfg.sh r6fg 14 tools/finish_ab.py 5 | grep -v amdgpu.ids | tail -16
