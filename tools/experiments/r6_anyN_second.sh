#!/bin/bash
# round 6: parity tests of the any-N spectral engine + kernel timeline of the cfg1 single-stage call
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -m gpu > gpurun_out/r6_anyN_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_anyN_tests.log
tail -15 gpurun_out/r6_anyN_tests.log
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py
