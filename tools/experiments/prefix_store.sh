#!/bin/bash
# round 4 (GPU box): what do the snapshot stores cost the prefix walk?  (variant_ps1: non-temporal stores, variant_ps2: no stores -- timing only)
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in libtspws_hip.so variant_ps3.so variant_ps2.so; do
  rm -rf /tmp/pp_$so
  TSPWS_JK_STAGES=1 TSPWS_LIB_PATH=$GRAFT_REPO_ROOT/ts-pws_amd/lib/$so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp_$so -o b -- python3 $GRAFT_REPO_ROOT/tools/cfg4_run.py > /tmp/pp_$so.log 2>&1
  printf "%-20s " $so; python3 $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pp_$so/b_results.db | grep k_prefix_walk
done
