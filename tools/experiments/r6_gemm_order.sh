#!/bin/bash
# round 6: where the matrix-pipe kernel of the clipped scales runs (before / behind the trace-lane kernel, third stream) + command-line batch tests and timing
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
for o in 0 1 2; do
  echo "== sweeps GEMM_ORDER=$o"
  TSPWS_LIB_PATH=$S TSPWS_GEMM_ORDER=$o python tools/cfg_bench.py cfg1 30
  TSPWS_LIB_PATH=$S TSPWS_GEMM_ORDER=$o TSPWS_SPEC_NSMAX=1024 python tools/cfg_bench.py cfg1 30
  TSPWS_LIB_PATH=$S TSPWS_GEMM_ORDER=$o python tools/cfg_bench.py c:500:20000 30
done
python tools/cfg_bench.py cfg2 40
} > gpurun_out/r6_gemm_order.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_gemm_order.log
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py | grep -v amdgpu.ids | tail -26
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_cli_sanitized_cpu.py -q -x > gpurun_out/r6_cli_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_cli_tests.log
tail -12 gpurun_out/r6_cli_tests.log
python tools/cli_timing.py 499 16501 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_cli_timing.txt
python tools/cli_timing.py 499 16501 TwoStage=10 unbiased 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6_cli_timing.txt
