#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_cycle.sh TAG [pytest-args]
# runs the GPU parity tests, then bench.py under rocprofv3 --kernel-trace --stats, into gpurun_out/
TAG=${1:-x}; shift
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -40) > gpurun_out/test_$TAG.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $R/gpurun_out/bench_$TAG.log 2>&1
cd $R
python $R/profiles/summarize_rocpd.py gpurun_out/prof_$TAG/b_results.db > gpurun_out/stats_$TAG.txt 2>&1
tail -15 gpurun_out/test_$TAG.log; grep -o '"value".\{0,40\}\|"ms_per_step".\{0,25\}\|"frac".\{0,25\}' gpurun_out/bench_$TAG.log; cat gpurun_out/stats_$TAG.txt
