#!/bin/bash
# round 4 (GPU box): the seeded random sweeps against the oracle on the rebuilt masked engine (new seeds), default switches and the snapshot form
cd $GRAFT_REPO_ROOT
python tools/random_sweep_features.py 9000 120 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
TSPWS_JK_DIRECT=0 TSPWS_JK_STAGES=3 python tools/random_sweep_features.py 9200 80 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep_jackknife.py 9400 40 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep.py 9500 120 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep_cli.py 9700 30 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
