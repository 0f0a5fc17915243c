cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30) > gpurun_out/test_r4a.log
bash tools/experiments/fwd_classes_cfg4.sh variant_abl.so > gpurun_out/fwd_classes_cfg4.txt 2>&1
python tools/cfg4_run.py > gpurun_out/cfg4_r4a.txt 2>&1
cat gpurun_out/test_r4a.log gpurun_out/fwd_classes_cfg4.txt; tail -1 gpurun_out/cfg4_r4a.txt
