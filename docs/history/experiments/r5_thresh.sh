#!/bin/bash
for sz in 32:32768 48:32768 32:131072 16:131072 64:8192 64:4096 128:4096 128:2048 256:2048 256:1024 512:1024 64:16384 40:16384 24:65536; do
  a=$(TSPWS_ENGINE=fir python3 tools/cfg_bench.py c:$sz 30 2>/dev/null | grep -o "[0-9.]* ms/call"); b=$(TSPWS_ENGINE=spectral python3 tools/cfg_bench.py c:$sz 30 2>/dev/null | grep -o "[0-9.]* ms/call")
  echo "$sz: fir $a | spectral $b"
done
