#!/bin/bash
# usage (GPU box): bash tools/variants.sh "args of fwd_bench.py" lib1.so lib2.so ...   -- the same timing for several builds, two rounds
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for so in "$@"; do
  printf "%-28s " "$(basename $so)"
  TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$so python3 $R/tools/fwd_bench.py $ARGS 2>&1 | grep -v amdgpu
done
done
