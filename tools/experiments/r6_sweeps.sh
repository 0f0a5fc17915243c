#!/bin/bash
# round 6: seeded sweeps against the oracle on the round's library -- any-N spectral engine + dense contraction (new seeds), then the older sweeps with new seeds
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
f() { grep -E "MISMATCH|cases|mismatch|Traceback|Error" | grep -v "waveletFamily"; }
{
export SWEEP_ANYN=1
TSPWS_ENGINE=spectral python tools/random_sweep_spectral.py 60000 60 2>&1 | f
python tools/random_sweep_spectral.py 60100 40 2>&1 | f
export TSPWS_LIB_PATH=$S
TSPWS_ENGINE=spectral TSPWS_SPEC_NT=min python tools/random_sweep_spectral.py 60200 40 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NT=double python tools/random_sweep_spectral.py 60300 40 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NT=min TSPWS_GEMM=0 python tools/random_sweep_spectral.py 60400 25 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NT=min TSPWS_GEMM_KS=3 TSPWS_GEMM_ORDER=0 python tools/random_sweep_spectral.py 60500 25 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NT=min TSPWS_GEMM_KS=200 TSPWS_GEMM_ORDER=2 TSPWS_SPEC_NSMAX=1048576 TSPWS_FEW_NSMAX=1048576 TSPWS_FEW_SPEC_MIN=8 python tools/random_sweep_spectral.py 60600 30 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=64 TSPWS_FEW_NSMAX=128 TSPWS_TL_BATCH=64 python tools/random_sweep_spectral.py 60700 25 2>&1 | f
TSPWS_ENGINE=spectral TSPWS_JK_EARLY_INV=0 python tools/random_sweep_spectral.py 60800 20 2>&1 | f
TSPWS_ENGINE=fir python tools/random_sweep_spectral.py 60900 10 2>&1 | f
unset SWEEP_ANYN TSPWS_LIB_PATH
TSPWS_ENGINE=spectral python tools/random_sweep_spectral.py 61000 40 2>&1 | f
python tools/random_sweep.py 62000 120 2>&1 | f
python tools/random_sweep_features.py 63000 60 2>&1 | f
python tools/random_sweep_jackknife.py 64000 40 2>&1 | f
python tools/random_sweep_masked.py 65000 60 2>&1 | f
python tools/random_sweep_device.py 66000 60 2>&1 | f
python tools/random_sweep_cli.py 67000 30 2>&1 | f
} 2>&1 | tee gpurun_out/r6_sweeps.txt
