#!/bin/bash
# round 4 (GPU box): the seeded random sweeps against the oracle on the rebuilt masked engine (new seeds), default switches and the snapshot form
cd $GRAFT_REPO_ROOT
python tools/random_sweep_features.py 9000 120 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
TSPWS_JK_DIRECT=0 TSPWS_JK_STAGES=3 python tools/random_sweep_features.py 9200 80 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep_jackknife.py 9400 40 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep.py 9500 120 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep_cli.py 9700 30 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -3
python tools/random_sweep_device.py 9800 60 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -2
python tools/random_sweep_large.py 9900 6 2>&1 | grep -v "Warning: Fold\|waveletFamily\|^tspws_main" | tail -2
# both instantiations of k_fwd_lds on every frame of the parity tests: 32 resident tap rows forced on the Morlet frames, 24 (tiled path) on the Mexican hat
for q in 24 32; do TSPWS_FWD_QT=$q timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "forward_inverse or golden or many_trace or jackknife or random_parameter or example" 2>&1 | grep "passed\|failed" | tail -1; done
