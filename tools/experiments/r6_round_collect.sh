#!/bin/bash
# round 6: the whole GPU suite on the final library, then everything the round's numbers come from (profiles/collect_round.sh r06)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r6_gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gpu_suite.log
tail -5 gpurun_out/r6_gpu_suite.log
bash profiles/collect_round.sh r06 > gpurun_out/r06_collect.log 2>&1
tail -c 6000 gpurun_out/r06_bench_line.json
cat gpurun_out/r06_stats.txt gpurun_out/r06_timeline.txt gpurun_out/r06_cfg1.txt
