#!/bin/bash
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
for sz in 256:32768 2048:32768 1024:8192 4096:8192 512:65536 256:131072 128:65536; do
  line="$sz:"
  for st in 12 96; do
    r=$(TSPWS_SPEC_TLSTEPS=$st python3 tools/cfg_bench.py c:$sz 30 2>/dev/null | grep -o "[0-9.]* ms/call"); line="$line steps $st: $r |"
  done
  echo "$line"
done
