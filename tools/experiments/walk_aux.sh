#!/bin/bash
# round 4 (GPU box): cache policy of the walking stream kernel's trace loads beside the forward transforms (cfg4, 2 stages)
cd $GRAFT_REPO_ROOT
for a in -1 2 0 18 19 17 3; do echo "aux $a"; for i in 1 2; do TSPWS_JK_STAGES=2 TSPWS_WALK_AUX=$a python tools/cfg4_run.py | tail -1; done; done
