#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
CFG4_REPS=3 bash tools/gpu_timeline_cfg.sh r6cfg4g 200 tools/cfg4_run.py > /dev/null 2>&1
grep -n "k_rows_walk\|k_epilogue\|copyBuffer\|k_seg_fix\|k_spec_transpose" gpurun_out/timeline_r6cfg4g.txt | head -60
