#!/bin/bash
# round 6: 64-point butterflies in the trace transform (one pass fewer at N >= 65536): parity + cfg4 / 256 x 131072 / 512 x 65536 A/B (TSPWS_SPEC_R64=0: the old plan)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -x 2>&1 | tail -3
{
for r in 1 0 1 0; do
  echo "== R64=$r"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r CFG4_REPS=30 python tools/cfg4_run.py
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:256:131072 30
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:512:65536 30
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:256:86400 30
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_r64.txt
bash tools/gpu_timeline_cfg.sh r6cfg4r64 30 tools/cfg4_run.py | grep -v amdgpu.ids | tail -24
