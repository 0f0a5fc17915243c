cd $GRAFT_REPO_ROOT
for w in 128 192 256; do for g in 1 2; do echo walk $w gps $g; TSPWS_JK_GPS=$g TSPWS_JK_WALK=$w python tools/cfg4_run.py | tail -1; done; done
