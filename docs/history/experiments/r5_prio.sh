#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
for pr in 0 1 -1; do
  for cfg in cfg2 cfg2d; do
    r=$(TSPWS_SIDE_PRIO=$pr python3 tools/cfg_bench.py $cfg 60 2>/dev/null | grep -o "[0-9.]* ms/call"); echo "side prio $pr $cfg: $r"
  done
done
