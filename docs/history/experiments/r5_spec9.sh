#!/bin/bash
mkdir -p gpurun_out
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1 TSPWS_SPEC_NSMAX=2048 TSPWS_SPEC_NSW=16 TSPWS_SPEC_NTB=2
for abl in 0 1 2 3 4 12 15; do
  export TSPWS_SPEC_ABL=$abl
  echo "== abl $abl"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "k_spec_fold"
done
