#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so TSPWS_ENGINE=spectral
for st in 1 2 4 8 96; do
  for cfg in cfg2 cfg2d; do
    r=$(TSPWS_SPEC_TLSTEPS=$st python3 tools/cfg_bench.py $cfg 60 2>/dev/null | grep -o "[0-9.]* ms/call, digest [0-9a-f]*"); echo "tlsteps $st $cfg: $r"
  done
done
