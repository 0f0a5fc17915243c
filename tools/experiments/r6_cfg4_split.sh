#!/bin/bash
# round 6: cfg4 -- which octaves of the 110 rows go through the spectral chain (TSPWS_FEW_NSMAX = largest N_s of the set; default N / 8 = 16384: D >= 8)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
CFG4_REPS=20 python tools/cfg4_run.py
for ns in 32768 16384 8192 4096 2048; do
  echo "== FEW_NSMAX=$ns"
  TSPWS_LIB_PATH=$S TSPWS_FEW_NSMAX=$ns CFG4_REPS=20 python tools/cfg4_run.py
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_cfg4_split.txt
