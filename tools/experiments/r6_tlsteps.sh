#!/bin/bash
# round 6: residue steps per trace-lane workgroup beside the spectral chain (TSPWS_SPEC_TLSTEPS, default 12: tuned on cfg2's 16 trace blocks) at 8-block batches
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for st in 12 6 24 48 96 12; do
  echo "== TLSTEPS=$st"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_TLSTEPS=$st python tools/cfg_bench.py cfg1 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_TLSTEPS=$st python tools/cfg_bench.py c:500:20000 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_TLSTEPS=$st python tools/cfg_bench.py c:256:16501 40
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_tlsteps.txt
