#!/bin/bash
# usage (GPU box): bash tools/experiments/fwd_steps.sh -- tap steps per wave of the direct kernel (more splits = shorter chains of dependent loads)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for st in 96 48 32 16; do
  echo "TSPWS_FWD_STEPS=$st"
  for s in "64 8192" "30 4096" "100 16384" "499 16501" "64 65536"; do TSPWS_FWD_STEPS=$st python3 $R/tools/small_run.py $s 2>&1 | grep " x "; done
  TSPWS_FWD_STEPS=$st python3 $R/tools/cfg_bench.py cfg3 20 2>&1 | grep -v amdgpu | tail -1
  TSPWS_FWD_STEPS=$st python3 $R/tools/cfg_bench.py cfg2 20 2>&1 | grep -v amdgpu | tail -1
done
