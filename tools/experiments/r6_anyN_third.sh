#!/bin/bash
# round 6: any-N spectral engine + matrix-pipe kernel for the clipped scales: parity + cfg1 timings + timeline
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -m gpu -x > gpurun_out/r6_anyN_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_anyN_tests.log
tail -8 gpurun_out/r6_anyN_tests.log
S=ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for e in fir auto; do
  echo "== TSPWS_ENGINE=$e"
  TSPWS_ENGINE=$e python tools/cfg_bench.py cfg1 20
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:500:20000 20
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:256:86400 10
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:400:5000 20
done
for nsmax in 512 1024 2048; do
  echo "== sweeps NSMAX=$nsmax"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$nsmax python tools/cfg_bench.py cfg1 20
done
for ks in 16 32 64 128; do
  echo "== sweeps GEMM_KS=$ks"
  TSPWS_LIB_PATH=$S TSPWS_GEMM_KS=$ks python tools/cfg_bench.py cfg1 20
done
echo "== sweeps GEMM=0"
TSPWS_LIB_PATH=$S TSPWS_GEMM=0 python tools/cfg_bench.py cfg1 20
echo "== sweeps NT=min / double at 5000, 20000"
for nt in min double; do
TSPWS_LIB_PATH=$S TSPWS_SPEC_NT=$nt python tools/cfg_bench.py c:400:5000 20
TSPWS_LIB_PATH=$S TSPWS_SPEC_NT=$nt python tools/cfg_bench.py c:500:20000 20
TSPWS_LIB_PATH=$S TSPWS_SPEC_NT=$nt python tools/cfg_bench.py c:300:30000 20
done
} > gpurun_out/r6_anyN_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_anyN_bench.log
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py | grep -v amdgpu.ids
