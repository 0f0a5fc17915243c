#!/bin/bash
# round 6: 1024-point inverse transforms of the chain as 64 x 16 (two passes, two levels) instead of 16 x 8 x 8 (TSPWS_SPEC_INV64=0: the old plan)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -x 2>&1 | tail -3
{
for r in 1 0 1 0; do
  echo "== INV64=$r"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_INV64=$r python tools/cfg_bench.py cfg1 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_INV64=$r python tools/cfg_bench.py cfg2 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_INV64=$r python tools/cfg_bench.py cfg2d 40
  TSPWS_LIB_PATH=$S TSPWS_SPEC_INV64=$r python tools/cfg_bench.py c:512:65536 30
  TSPWS_LIB_PATH=$S TSPWS_SPEC_INV64=$r python tools/cfg_bench.py c:256:131072 20
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_inv64.txt
