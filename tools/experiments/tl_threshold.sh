#!/bin/bash
# usage (GPU box): bash tools/experiments/tl_threshold.sh -- few-trace kernels vs trace-lane kernel around tspws_many_trace_path's threshold
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for shape in 64:32768 100:8192 128:16384 128:65536 160:32768 192:16384 256:8192 256:16384 320:8192 499:16501 512:4096 1024:4096; do
  for mode in 1000000 1; do
    printf "%-12s TL_MIN=%-8s " $shape $mode
    TSPWS_TL_MIN=$mode python3 $R/tools/cfg_bench.py c:$shape 30 2>&1 | grep -v amdgpu | tail -1
  done
done
