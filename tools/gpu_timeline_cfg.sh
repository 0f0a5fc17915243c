#!/bin/bash
# usage (GPU box): bash tools/gpu_timeline_cfg.sh TAG NKERNELS script.py [args] -- rocprofv3 kernel trace of a tools/ script: stats + timeline of its last kernels
TAG=$1; NK=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/$1 "${@:2}" > $R/gpurun_out/run_$TAG.log 2>&1
cd $R
python $R/profiles/summarize_rocpd.py gpurun_out/prof_$TAG/b_results.db > gpurun_out/stats_$TAG.txt 2>&1
python $R/profiles/timeline_rocpd.py gpurun_out/prof_$TAG/b_results.db $NK > gpurun_out/timeline_$TAG.txt 2>&1
rm -rf gpurun_out/prof_$TAG
tail -3 gpurun_out/run_$TAG.log; cat gpurun_out/stats_$TAG.txt gpurun_out/timeline_$TAG.txt
