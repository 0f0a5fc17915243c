#!/bin/bash
# round 5, first GPU call: read ceiling (same box as a bench line), cfg1 single-stage kernel stats / timeline / VALU mix
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
./tools/read_ceiling > gpurun_out/r05_read_ceiling_raw.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r05a_bench_line.json 2> gpurun_out/r05a_bench.err
./tools/read_ceiling >> gpurun_out/r05_read_ceiling_raw.txt 2>&1
bash tools/gpu_prof_cfg.sh r05cfg1 tools/cfg1s_run.py > /dev/null 2>&1
bash tools/gpu_timeline_cfg.sh r05cfg1t 30 tools/cfg1s_run.py > /dev/null 2>&1
bash tools/experiments/pmc_valu_mix.sh r05cfg1 tools/cfg1s_run.py > /dev/null 2>&1
python3 tools/cfg1_run.py 2>/dev/null | grep cfg1 > gpurun_out/r05a_cfg1.txt
cat gpurun_out/r05_read_ceiling_raw.txt; tail -c 1500 gpurun_out/r05a_bench_line.json; cat gpurun_out/stats_r05cfg1.txt gpurun_out/timeline_r05cfg1t.txt gpurun_out/r05a_cfg1.txt; head -80 gpurun_out/pmcv_r05cfg1.txt
