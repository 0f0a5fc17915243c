#!/bin/bash
# k_rows_walk: W <= 12 instantiation with 8 / 16 loads in flight against the 16-column form (cfg4), + upload probe
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
mkdir -p gpurun_out
for nl in 8 16; do
  export TSPWS_WALK_NL=$nl
  echo "== W12, NL $nl"; bash tools/gpu_prof_cfg.sh r05w tools/cfg4_run.py 2>&1 | grep "cfg4\|k_rows_walk\|k_fwd_lds\|k_seg_fix"
done
unset TSPWS_LIB_PATH TSPWS_WALK_NL
timeout 600 python3 tools/probes/upload_probe.py 2>&1 | tail -12
