#!/bin/bash
# round 4 (GPU box): k_rows_walk by loads in flight per thread (ROWS_NB build variants), one stage (no overlap)
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  rm -rf /tmp/rw_$so
  TSPWS_JK_STAGES=1 TSPWS_LIB_PATH=$GRAFT_REPO_ROOT/ts-pws_amd/lib/$so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rw_$so -o b -- python3 $GRAFT_REPO_ROOT/tools/cfg4_run.py > /tmp/rw_$so.log 2>&1
  printf "%-20s " $so; python3 $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/rw_$so/b_results.db | grep k_rows_walk\|k_add_halves; tail -1 /tmp/rw_$so.log
done
