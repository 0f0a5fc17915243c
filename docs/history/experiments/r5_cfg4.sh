#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spectral_gpu.py -q -k "jackknife_rows" 2>&1 | tail -5
for e in fir auto; do
  if [ $e = fir ]; then export TSPWS_ENGINE=fir; else unset TSPWS_ENGINE; fi
  echo "== engine $e"; bash tools/gpu_timeline_cfg.sh r05c4$e 26 tools/cfg4_run.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -48
done
