#!/bin/bash
for e in fir auto fir auto; do
  if [ $e = fir ]; then export TSPWS_ENGINE=fir; else unset TSPWS_ENGINE; fi
  echo "== engine $e"; python3 tools/cfg4_run.py 2>&1 | grep -v amdgpu | tail -3
done
