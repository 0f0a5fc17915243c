#!/bin/bash
# usage (GPU box): bash tools/experiments/tl_pick.sh -- the two decompositions of the many-trace path (TlTable 0: octaves with >= 33 outputs on k_fwd_tl, 1: >= 257) per shape
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for shape in 128:65536 256:32768 499:16501 512:16384 768:16384 1024:8192 2048:8192 cfg2; do
  for pick in 0 1; do
    printf "%-12s PICK=%s " $shape $pick
    s=$shape; [ $shape = cfg2 ] || s=c:$shape
    TSPWS_TL_PICK=$pick python3 $R/tools/cfg_bench.py $s 30 2>&1 | grep -v amdgpu | tail -1
  done
done
