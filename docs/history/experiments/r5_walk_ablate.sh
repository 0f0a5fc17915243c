#!/bin/bash
# k_rows_walk alone under rocprofv3: the shipped form, the writer wave storing nothing (ROWS_ABL=1), no flush at all (ROWS_ABL=2) -- results wrong, clock only
R=${GRAFT_REPO_ROOT:-$PWD}
for v in libtspws_hip variant_ra1 variant_ra2; do
  TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$v.so bash tools/gpu_timeline_cfg.sh wab 200 tools/experiments/r5_walk_alone.py > /dev/null 2>&1
  echo "$v: $(grep k_rows_walk gpurun_out/timeline_wab.txt | awk '{s+=$3; n++} END {printf "%.1f us (mean of %d)", s/n, n}')"
done
