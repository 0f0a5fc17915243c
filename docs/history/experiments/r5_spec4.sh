#!/bin/bash
mkdir -p gpurun_out
export TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=2048 TSPWS_SPEC_SERIAL=1
for nsw in 24 48; do
  export TSPWS_SPEC_NSW=$nsw TSPWS_SPEC_NTB=1
  for ntr in 1024 256; do
  echo "== nsw $nsw ntr $ntr"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py $ntr 2>&1 | grep "cfg2 ms\|k_spec\|k_fwd_tl\|k_transpose"
  done
done
