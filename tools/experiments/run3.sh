cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q -x -k "jackknife or subsampl or golden or random_parameter" 2>&1 | tail -8) > gpurun_out/test_r4b.log
cat gpurun_out/test_r4b.log
for g in 1 2; do echo gps $g; TSPWS_JK_GPS=$g python tools/cfg4_run.py | tail -1; done
TSPWS_JK_PIPELINE=0 python tools/cfg4_run.py | tail -1
bash tools/gpu_timeline_cfg.sh r4b_cfg4 60 tools/cfg4_run.py | tail -64
