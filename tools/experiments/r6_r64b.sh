#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for r in 1 0 1 0; do
  echo "== R64=$r"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:512:65536 30
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:2048:65536 20
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:256:65536 30
  TSPWS_LIB_PATH=$S TSPWS_SPEC_R64=$r python tools/cfg_bench.py c:500:40000 30
done
CFG4_REPS=30 python tools/cfg4_run.py
python tools/cfg_bench.py cfg1 40
python tools/cfg_bench.py cfg2 40
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_r64b.txt
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -x 2>&1 | tail -3
