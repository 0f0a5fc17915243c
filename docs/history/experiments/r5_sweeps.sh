#!/bin/bash
# round 5: seeded sweeps of the spectral engine against the oracle (new seeds per setting), the engine pinned per process
R=$(pwd); export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/libtspws_hip_sweeps.so
f() { grep -i "mismatch" | tail -3; }
TSPWS_ENGINE=spectral python3 tools/random_sweep_spectral.py 20000 40 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=1048576 TSPWS_FEW_NSMAX=1048576 TSPWS_FEW_SPEC_MIN=8 python3 tools/random_sweep_spectral.py 21000 40 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=64 TSPWS_FEW_NSMAX=128 TSPWS_FEW_SPEC_MIN=20 python3 tools/random_sweep_spectral.py 22000 30 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=512 TSPWS_SPEC_FOLD=lds python3 tools/random_sweep_spectral.py 23000 20 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=4096 TSPWS_SPEC_FOLD=lds TSPWS_SPEC_NSW=8 TSPWS_SPEC_NTB=1 python3 tools/random_sweep_spectral.py 24000 15 2>&1 | f
unset TSPWS_LIB_PATH
python3 tools/random_sweep_spectral.py 25000 25 2>&1 | f
TSPWS_ENGINE=fir python3 tools/random_sweep_spectral.py 26000 10 2>&1 | f
