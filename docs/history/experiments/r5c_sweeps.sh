#!/bin/bash
# round 5, after the walk / transposition / fold changes of the second half: the masked, jackknife, device and spectral sweeps with new seeds
cd ${GRAFT_REPO_ROOT:-.}
f() { grep -i "mismatch" | tail -1; }
python tools/random_sweep_jackknife.py 16400 40 2>&1 | f
python tools/random_sweep_masked.py 16600 60 2>&1 | f
python tools/random_sweep_device.py 16800 60 2>&1 | f
python tools/random_sweep_features.py 16000 60 2>&1 | f
python tools/random_sweep_spectral.py 16900 40 2>&1 | f
python tools/random_sweep_cli.py 16700 12 2>&1 | f
export TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 python tools/random_sweep_masked.py 17000 40 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 TSPWS_FEW_NSMAX=64 python tools/random_sweep_jackknife.py 17100 30 2>&1 | f
TSPWS_JK_STAGES=3 python tools/random_sweep_masked.py 17200 30 2>&1 | f
TSPWS_JK_DIRECT=0 python tools/random_sweep_masked.py 17300 20 2>&1 | f
