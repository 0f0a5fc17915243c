#!/bin/bash
# round 6: octave bound of the spectral set on many-trace batches, re-swept on the round's library (TSPWS_SPEC_NSMAX = largest N_s of the set)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for ns in 512 1024 2048 4096; do
  echo "== NSMAX=$ns"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$ns python tools/cfg_bench.py cfg2 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$ns python tools/cfg_bench.py cfg2d 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$ns python tools/cfg_bench.py c:4096:8192 40
done
for ns in 2048 4096 8192 16384; do
  echo "== NSMAX=$ns (256 x 131072)"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$ns python tools/cfg_bench.py c:256:131072 20
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_cfg2_bound.txt
