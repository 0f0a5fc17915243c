#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q 2>&1 | tail -3
echo "== size sweep: fir | spectral nsmax"
for sz in 256:32768 1024:32768 2048:32768 1024:8192 4096:8192 2048:16384 512:65536 1024:131072; do
  line="$sz:"
  r=$(TSPWS_ENGINE=fir python3 tools/cfg_bench.py c:$sz 5 2>/dev/null | grep -o "[0-9.]* ms/call"); line="$line fir $r |"
  for ns in 512 1024 2048 4096 8192; do
    r=$(TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=$ns python3 tools/cfg_bench.py c:$sz 5 2>/dev/null | grep -o "[0-9.]* ms/call"); line="$line $ns: $r |"
  done
  echo "$line"
done
