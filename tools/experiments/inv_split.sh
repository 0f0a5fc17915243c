#!/bin/bash
# usage (GPU box): bash tools/experiments/inv_split.sh -- the inverse with octave items (0) against one item per scale (1), per frame length
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for split in 0 1 0 1; do
  echo "TSPWS_INV_SPLIT=$split"
  TSPWS_INV_SPLIT=$split python3 $R/tools/cfg1_run.py 2>&1 | grep cfg1
  for s in c:64:8192 c:100:32768 c:64:65536 cfg2 cfg3; do TSPWS_INV_SPLIT=$split python3 $R/tools/cfg_bench.py $s 30 2>&1 | grep -v amdgpu | tail -1; done
done
