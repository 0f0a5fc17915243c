#!/bin/bash
# usage: bash gpu_prof.sh TAG [ENV=VAL ...] -- profile bench.py (no tests) with optional env
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $R/gpurun_out/bench_$TAG.log 2>&1
cd $R
python $R/profiles/summarize_rocpd.py gpurun_out/prof_$TAG/b_results.db > gpurun_out/stats_$TAG.txt 2>&1
grep -o '"value".\{0,40\}\|"ms_per_step".\{0,25\}\|"frac".\{0,25\}' gpurun_out/bench_$TAG.log; cat gpurun_out/stats_$TAG.txt
