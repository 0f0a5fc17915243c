#!/bin/bash
# usage (GPU box, repo root): bash tools/sweep_env.sh TAG "ENV=VAL ENV2=VAL" "ENV=VAL" ...
# Runs bench.py (no CPU leg, no profiler) once per environment set, all in ONE call so that the numbers share a box
# (box-to-box spread of the streaming kernel is a few per cent), and prints ms_per_step + the streaming-stage time.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/sweep_$TAG.txt
: > $OUT
for rep in 1 2; do
for envs in "$@"; do
	line=$(env $envs timeout 300 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu ${BENCH_ARGS} 2>/dev/null | tail -1)
	ms=$(echo "$line" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
	st=$(echo "$line" | grep -o '"ms_per_launch": [0-9.]*' | cut -d' ' -f2)
	printf "%-60s ms_per_step %s  streaming %s\n" "[$envs]" "$ms" "$st" | tee -a $OUT
done
done
