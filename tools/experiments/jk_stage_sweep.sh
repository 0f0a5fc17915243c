#!/bin/bash
# round 4 (GPU box): the pipelined masked-replica call by number of stages (TSPWS_JK_STAGES), and the serial call (TSPWS_JK_PIPELINE=0)
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q -x -k "jackknife or subsampl or golden or random_parameter" 2>&1 | grep -v "Warning: Fold\|waveletFamily" | tail -3)
for n in 1 2 3 4 5 10; do echo stages $n; TSPWS_JK_STAGES=$n python tools/cfg4_run.py | tail -1; done
echo serial; TSPWS_JK_PIPELINE=0 python tools/cfg4_run.py | tail -1
bash tools/gpu_timeline_cfg.sh r4h_cfg4 22 tools/cfg4_run.py | tail -23
