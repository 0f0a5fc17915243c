#!/bin/bash
# round 6: matrix-pipe kernel with its own run reduction: cfg1 timings, run-count sweep, octave bound, timelines side by side / one after the other
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=ts-pws_amd/lib/libtspws_hip_sweeps.so
{
python tools/cfg_bench.py cfg1 20
for ks in 16 32 64 128; do
  echo "== sweeps GEMM_KS=$ks"
  TSPWS_LIB_PATH=$S TSPWS_GEMM_KS=$ks python tools/cfg_bench.py cfg1 20
done
for nsmax in 512 1024 2048; do
  echo "== sweeps NSMAX=$nsmax"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$nsmax python tools/cfg_bench.py cfg1 20
done
} > gpurun_out/r6_anyN_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_anyN_bench.log
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py | grep -v amdgpu.ids | tail -26
TSPWS_LIB_PATH=$S TSPWS_SPEC_SERIAL=1 bash tools/gpu_timeline_cfg.sh r6cfg1ser 24 tools/cfg1s_run.py | grep -v amdgpu.ids | tail -26
