#!/bin/bash
# round 6, first run of the spectral engine for any N (periodic extension): parity tests of the engine + cfg1 timings FIR / default / forced geometries
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_spectral_gpu.py -x -q -m gpu > gpurun_out/r6_anyN_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_anyN_tests.log
tail -15 gpurun_out/r6_anyN_tests.log
{
for e in fir auto spectral; do
  echo "== TSPWS_ENGINE=$e"
  TSPWS_ENGINE=$e python tools/cfg_bench.py cfg1 20
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:499:16384 20
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:500:20000 20
  TSPWS_ENGINE=$e python tools/cfg_bench.py c:256:86400 10
done
S=ts-pws_amd/lib/libtspws_hip_sweeps.so
for nsmax in 256 512 1024 2048 4096; do
  echo "== sweeps NSMAX=$nsmax"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$nsmax python tools/cfg_bench.py cfg1 20
done
echo "== sweeps NT=double"
TSPWS_LIB_PATH=$S TSPWS_SPEC_NT=double python tools/cfg_bench.py cfg1 20
TSPWS_LIB_PATH=$S TSPWS_SPEC_NT=double TSPWS_SPEC_NSMAX=1024 python tools/cfg_bench.py cfg1 20
echo "== serial chain"
TSPWS_LIB_PATH=$S TSPWS_SPEC_SERIAL=1 python tools/cfg_bench.py cfg1 20
} > gpurun_out/r6_anyN_bench.log 2>&1
cat gpurun_out/r6_anyN_bench.log
