#!/bin/bash
# round 6: the few-rows octave bound again, now that the inverses of the FIR octaves run beside the chain's tail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for ns in 8192 4096 8192 4096 16384; do
  echo "== FEW_NSMAX=$ns"
  TSPWS_LIB_PATH=$S TSPWS_FEW_NSMAX=$ns CFG4_REPS=30 python tools/cfg4_run.py
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_cfg4_split2.txt
