#!/bin/bash
# usage (GPU box, repo root): bash profiles/collect_round.sh rNN
# Everything the round's numbers come from, into gpurun_out/ (copy the summaries you keep into profiles/):
#   bench line with the CPU baseline, rocprofv3 kernel stats + timeline of the same command, PMC traffic passes.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_line.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/${TAG}_bench_prof.log 2>&1
cd $R
python3 profiles/summarize_rocpd.py gpurun_out/prof_$TAG/b_results.db > gpurun_out/${TAG}_stats.txt 2>&1
python3 profiles/timeline_rocpd.py gpurun_out/prof_$TAG/b_results.db 14 > gpurun_out/${TAG}_timeline.txt 2>&1
bash profiles/collect_pmc.sh $TAG > /dev/null 2>&1
# the traffic record bench.py quotes: counters of THIS build's streaming kernel, tagged with the hash of its source
python3 profiles/pmc_record.py gpurun_out/pmc_$TAG > gpurun_out/pmc_partial_stacks.json 2> gpurun_out/${TAG}_pmc_record.err
python3 tools/cfg1_run.py 2>/dev/null | grep cfg1 > gpurun_out/${TAG}_cfg1.txt
cp gpurun_out/pmc_partial_stacks.json profiles/pmc_partial_stacks.json 2>/dev/null  # (bench.py quotes it from profiles/: hash-guarded)
# kernel stats of the other configs (bench.py's other_configs leg times them; these are the per-kernel breakdowns)
bash tools/gpu_prof_cfg.sh ${TAG}cfg2 tools/cfg2_run.py > /dev/null 2>&1; cp gpurun_out/stats_${TAG}cfg2.txt gpurun_out/${TAG}_stats_cfg2.txt
bash tools/gpu_timeline_cfg.sh ${TAG}cfg4 70 tools/cfg4_run.py > /dev/null 2>&1; cp gpurun_out/stats_${TAG}cfg4.txt gpurun_out/${TAG}_stats_cfg4.txt; cp gpurun_out/timeline_${TAG}cfg4.txt gpurun_out/${TAG}_timeline_cfg4.txt
# counters of the forward kernel on the Mexican-hat frame (VALU vs FMA instruction counts: separate --pmc passes)
PMC_CMD="tools/cfg4_run.py" PMC_SETS="sq sq2" bash profiles/collect_pmc.sh ${TAG}cfg4 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}cfg4.txt gpurun_out/${TAG}_pmc_cfg4.txt
# whole-call HBM traffic of cfg4 (FETCH_SIZE / WRITE_SIZE passes over every kernel of the call): the hash-guarded record bench.py quotes as cfg4's roofline.traffic
PMC_CMD="tools/cfg4_run.py" PMC_SETS="fetch write" bash profiles/collect_pmc.sh ${TAG}cfg4hbm > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}cfg4hbm.txt gpurun_out/${TAG}_pmc_cfg4_hbm.txt
python3 profiles/pmc_record_call.py gpurun_out/pmc_${TAG}cfg4hbm k_rows_walk > gpurun_out/pmc_cfg4_call.json 2> gpurun_out/${TAG}_pmc_record_call.err && cp gpurun_out/pmc_cfg4_call.json profiles/pmc_cfg4_call.json
# the spectral chain of cfg2 on the SHIPPED library (k_spec_fold_mfma: matrix-pipe busy cycles, VALU mix, fetched / written bytes)
PMC_CMD="tools/cfg2_run.py" PMC_SETS="sq mfma fetch write" bash profiles/collect_pmc.sh ${TAG}spectral > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}spectral.txt gpurun_out/${TAG}_pmc_spectral.txt
# the shipped example's shape (499 x 16501 single-stage): kernel stats + timeline + the matrix-pipe / VALU counters of its kernels
bash tools/gpu_timeline_cfg.sh ${TAG}cfg1 26 tools/cfg1s_run.py > /dev/null 2>&1; cp gpurun_out/stats_${TAG}cfg1.txt gpurun_out/${TAG}_stats_cfg1.txt; cp gpurun_out/timeline_${TAG}cfg1.txt gpurun_out/${TAG}_timeline_cfg1.txt
PMC_CMD="tools/cfg1s_run.py" PMC_SETS="sq mfma fetch write" bash profiles/collect_pmc.sh ${TAG}cfg1 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}cfg1.txt gpurun_out/${TAG}_pmc_cfg1.txt
rm -rf gpurun_out/prof_${TAG}* gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}cfg4 gpurun_out/pmc_${TAG}cfg4hbm gpurun_out/pmc_${TAG}spectral gpurun_out/pmc_${TAG}cfg1
tail -c 3500 gpurun_out/${TAG}_bench_line.json; cat gpurun_out/${TAG}_stats.txt gpurun_out/${TAG}_timeline.txt gpurun_out/${TAG}_cfg1.txt gpurun_out/${TAG}_stats_cfg2.txt gpurun_out/${TAG}_stats_cfg4.txt; grep -A30 "k_partial" gpurun_out/pmc_$TAG.txt | head -40
