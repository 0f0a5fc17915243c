#!/bin/bash
# usage (GPU box): bash tools/variants_cmd.sh "tools/script.py args" lib1.so lib2.so ...  -- the same script under several builds, two rounds
CMD=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for so in "$@"; do
  printf "%-28s " "$(basename $so)"
  TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$so python3 $R/$CMD 2>&1 | grep -v amdgpu | tail -1
done
done
