#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so TSPWS_ENGINE=spectral
for ns in 1024 2048 4096; do
  for cfg in cfg2 cfg2d; do
    r=$(TSPWS_SPEC_NSMAX=$ns python3 tools/cfg_bench.py $cfg 60 2>/dev/null | grep -o "[0-9.]* ms/call"); echo "nsmax $ns $cfg: $r"
  done
done
