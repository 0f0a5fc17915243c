#!/bin/bash
# spectral engine, first run: parity check, then cfg2 timing per NSMAX with kernel stats
mkdir -p gpurun_out
timeout 900 python3 tools/spectral_check.py quick > gpurun_out/spec_check.log 2>&1
tail -30 gpurun_out/spec_check.log
for e in fir spectral; do
  for ns in 256 1024 4096; do
    [ $e = fir ] && [ $ns != 256 ] && continue
    echo "== engine $e nsmax $ns"; TSPWS_ENGINE=$e TSPWS_SPEC_NSMAX=$ns timeout 300 python3 tools/cfg2_run.py 2>&1 | tail -2
  done
done
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=1024 bash tools/gpu_timeline_cfg.sh r05spec 24 tools/cfg2_run.py 2>&1 | tail -60
