#!/bin/bash
# round 5, final library: the seeded sweeps of rounds 2-4 with NEW seeds (shipped library: default engine rule), then two settings on the sweeps build
cd ${GRAFT_REPO_ROOT:-.}
f() { grep -i "mismatch" | tail -1; }
python tools/random_sweep_features.py 14000 80 2>&1 | f
python tools/random_sweep.py 14500 120 2>&1 | f
python tools/random_sweep_jackknife.py 14400 30 2>&1 | f
python tools/random_sweep_masked.py 14600 40 2>&1 | f
python tools/random_sweep_cli.py 14700 20 2>&1 | f
python tools/random_sweep_device.py 14800 60 2>&1 | f
python tools/random_sweep_large.py 14900 6 2>&1 | f
export TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 python tools/random_sweep_masked.py 15000 30 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_FEW_SPEC_MIN=12 TSPWS_FEW_NSMAX=64 python tools/random_sweep_jackknife.py 15100 20 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=100000 python tools/random_sweep_features.py 15200 30 2>&1 | f
