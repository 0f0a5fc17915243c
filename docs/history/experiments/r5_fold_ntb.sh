#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1
for ns in 1024 2048; do for ntb in 2 1; do
  export TSPWS_SPEC_NSMAX=$ns TSPWS_SPEC_NTB=$ntb
  echo "== nsmax $ns ntb $ntb"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "k_spec_fold\|cfg2 ms"
done; done
