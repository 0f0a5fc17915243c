#!/bin/bash
# round 5, FINAL library: every seeded sweep once more with new seeds and larger counts (shipped library), then the forced paths on the sweeps build
cd ${GRAFT_REPO_ROOT:-.}
f() { grep -i "mismatch" | tail -1; }
python tools/random_sweep.py 21000 200 2>&1 | f
python tools/random_sweep_features.py 21100 120 2>&1 | f
python tools/random_sweep_jackknife.py 21200 80 2>&1 | f
python tools/random_sweep_masked.py 21300 120 2>&1 | f
python tools/random_sweep_device.py 21400 80 2>&1 | f
python tools/random_sweep_spectral.py 21500 80 2>&1 | f
python tools/random_sweep_cli.py 21600 20 2>&1 | f
python tools/random_sweep_large.py 21700 6 2>&1 | f
export TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 python tools/random_sweep_masked.py 21800 60 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 TSPWS_FEW_NSMAX=64 python tools/random_sweep_jackknife.py 21900 40 2>&1 | f
TSPWS_SPEC_PARALLEL=0 TSPWS_FEW_SPEC_MIN=12 python tools/random_sweep_masked.py 22000 40 2>&1 | f
TSPWS_JK_STAGES=2 TSPWS_FEW_SPEC_MIN=12 python tools/random_sweep_masked.py 22100 40 2>&1 | f
TSPWS_JK_DIRECT=0 python tools/random_sweep_jackknife.py 22200 30 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=100000 python tools/random_sweep_features.py 22300 40 2>&1 | f
TSPWS_ENGINE=fir python tools/random_sweep.py 22400 60 2>&1 | f
