cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q -x -k "jackknife or subsampl or golden or random_parameter or convergence or partial" 2>&1 | tail -4) 
for w in 0 256 512; do echo walk $w; TSPWS_JK_WALK=$w python tools/cfg4_run.py | tail -1; done
TSPWS_JK_PIPELINE=0 python tools/cfg4_run.py | tail -1
TSPWS_JK_WALK=256 bash tools/gpu_timeline_cfg.sh r4e_cfg4 30 tools/cfg4_run.py | tail -31
