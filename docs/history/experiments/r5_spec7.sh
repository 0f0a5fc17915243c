#!/bin/bash
mkdir -p gpurun_out
timeout 900 python3 tools/spectral_check.py quick > gpurun_out/spec_check.log 2>&1
grep -c "e-1[0-9]\|0.00e+00" gpurun_out/spec_check.log; grep -i "error\|assert\|Traceback" gpurun_out/spec_check.log | head
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1 TSPWS_SPEC_NSMAX=2048
for nsw in 24 16; do for ntb in 1 2; do
  [ $nsw = 24 ] && [ $ntb = 2 ] && continue
  export TSPWS_SPEC_NSW=$nsw TSPWS_SPEC_NTB=$ntb
  echo "== nsw $nsw ntb $ntb"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "k_spec_fold"
done; done
