#!/bin/bash
mkdir -p gpurun_out
timeout 900 python3 tools/spectral_check.py quick > gpurun_out/spec_check.log 2>&1
grep -c "e-1[0-9]\|0.00e+00" gpurun_out/spec_check.log; grep -i "error\|assert\|Traceback" gpurun_out/spec_check.log | head
for ns in 512 1024 2048 4096 8192; do
  for ser in 0 1; do
    echo "== spectral nsmax $ns serial $ser"; if [ $ser = 1 ]; then export TSPWS_SPEC_SERIAL=1; else unset TSPWS_SPEC_SERIAL; fi
    TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=$ns timeout 300 python3 tools/cfg2_run.py 2>&1 | tail -1
  done
done
unset TSPWS_SPEC_SERIAL
echo "== fir"; TSPWS_ENGINE=fir timeout 300 python3 tools/cfg2_run.py 2>&1 | tail -1
TSPWS_SPEC_SERIAL=1 TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=2048 bash tools/gpu_timeline_cfg.sh r05spec 16 tools/cfg2_run.py 2>&1 | tail -45
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=2048 bash tools/gpu_timeline_cfg.sh r05specb 16 tools/cfg2_run.py 2>&1 | tail -18
