#!/bin/bash
mkdir -p gpurun_out
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1 TSPWS_SPEC_NSMAX=2048
for nsw in 16 8; do for ntb in 1 2; do
  export TSPWS_SPEC_NSW=$nsw TSPWS_SPEC_NTB=$ntb
  echo "== nsw $nsw ntb $ntb"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "k_spec_fold"
done; done
