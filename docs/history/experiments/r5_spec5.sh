#!/bin/bash
mkdir -p gpurun_out
timeout 900 python3 tools/spectral_check.py quick > gpurun_out/spec_check.log 2>&1
grep -c "e-1[0-9]\|0.00e+00" gpurun_out/spec_check.log; grep -i "error\|assert\|Traceback" gpurun_out/spec_check.log | head
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1
for ns in 2048 4096; do
  export TSPWS_SPEC_NSMAX=$ns
  echo "== nsmax $ns serial"; bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "cfg2 ms\|k_spec\|k_fwd_tl\|k_transpose"
done
unset TSPWS_SPEC_SERIAL
for ns in 1024 2048 4096; do echo "== nsmax $ns overlapped"; TSPWS_SPEC_NSMAX=$ns python3 tools/cfg2_run.py 2>&1 | grep "cfg2 ms"; done
