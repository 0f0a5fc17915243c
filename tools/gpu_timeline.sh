#!/bin/bash
# usage (GPU box): bash tools/gpu_timeline.sh TAG [ENV=VAL ...] -- rocprofv3 kernel trace of bench.py, timeline of the last step
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $R/gpurun_out/bench_$TAG.log 2>&1
cd $R
grep -o '"ms_per_step".\{0,25\}' gpurun_out/bench_$TAG.log
python $R/profiles/timeline_rocpd.py gpurun_out/prof_$TAG/b_results.db 12 | tee gpurun_out/timeline_$TAG.txt
