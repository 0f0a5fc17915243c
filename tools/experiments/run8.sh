cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | grep -v "Warning: Fold\|waveletFamily" | tail -15) > gpurun_out/test_r4c.log
cat gpurun_out/test_r4c.log
python tools/cfg4_run.py | tail -1
