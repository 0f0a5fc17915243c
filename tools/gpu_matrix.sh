#!/bin/bash
# clean (un-profiled) timing matrix of the whole call
for cfg in "TSPWS_OVERLAP=0 TSPWS_FWD_NOLDS=1" "TSPWS_OVERLAP=0 TSPWS_FWD_NOLDS=0" \
           "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=2 TSPWS_FWD_NOLDS=1" "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=5 TSPWS_FWD_NOLDS=1" "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=10 TSPWS_FWD_NOLDS=1" \
           "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=2 TSPWS_FWD_NOLDS=0" "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=5 TSPWS_FWD_NOLDS=0" "TSPWS_OVERLAP=1 TSPWS_PIPE_BATCH=10 TSPWS_FWD_NOLDS=0" $EXTRA; do
  r=$(env $cfg python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')
  echo "$cfg -> $r"
done
