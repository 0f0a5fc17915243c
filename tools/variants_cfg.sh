#!/bin/bash
# usage (GPU box): bash tools/variants_cfg.sh "args of cfg_bench.py" lib1.so lib2.so ...
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for so in "$@"; do
  printf "%-28s " "$(basename $so)"
  TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$so python3 $R/tools/cfg_bench.py $ARGS 2>&1 | grep -v amdgpu
done
done
