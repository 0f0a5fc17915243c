#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
for sz in 1024:32768:mexhat 4096:8192:mexhat 512:65536:mexhat; do
  line="$sz:"
  r=$(TSPWS_ENGINE=fir python3 tools/cfg_bench.py c:$sz 20 2>/dev/null | grep -o "[0-9.]* ms/call, digest [0-9a-f]*"); line="$line fir $r |"
  for ns in 512 1024 2048 4096; do
    r=$(TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=$ns python3 tools/cfg_bench.py c:$sz 20 2>/dev/null | grep -o "[0-9.]* ms/call, digest [0-9a-f]*"); line="$line $ns: $r |"
  done
  echo "$line"
done
