cd $GRAFT_REPO_ROOT
for w in 0 256 512 1024 2048; do echo walk $w; TSPWS_JK_WALK=$w python tools/cfg4_run.py | tail -1; done
TSPWS_JK_WALK=512 TSPWS_JK_GPS=2 python tools/cfg4_run.py | tail -1
bash tools/gpu_timeline_cfg.sh r4c_cfg4 60 tools/cfg4_run.py | tail -64
