#!/bin/bash
# usage (GPU box): bash tools/experiments/fuse_slices.sh -- workgroups the fused forward launch of a SHORT frame aims at (TSPWS_FUSE_WGS; 1 = the old 32-trace slices)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for wgs in 1 256 512 768 1024 1536; do
  echo "TSPWS_FUSE_WGS=$wgs"
  TSPWS_FUSE_WGS=$wgs python3 $R/tools/small_run.py 64 8192 2>&1 | grep " x "
  TSPWS_FUSE_WGS=$wgs python3 $R/tools/small_run.py 30 4096 2>&1 | grep " x "
  TSPWS_FUSE_WGS=$wgs python3 $R/tools/small_run.py 100 16384 2>&1 | grep " x "
  TSPWS_FUSE_WGS=$wgs python3 $R/tools/small_run.py 499 16501 2>&1 | grep " x "
  TSPWS_FUSE_WGS=$wgs python3 $R/tools/small_run.py 64 65536 2>&1 | grep " x "
done
