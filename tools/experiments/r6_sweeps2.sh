#!/bin/bash
# round 6, after the 64-point first pass: any-N sweeps whose windows are 65 536 (trace lengths 16 385 .. 40 000, >= 193 traces in most cases), default and forced-double windows
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
f() { grep -E "MISMATCH|cases|mismatch|Traceback|Error" | grep -v "waveletFamily"; }
{
export SWEEP_ANYN=1
TSPWS_ENGINE=spectral python tools/random_sweep_spectral.py 70000 50 2>&1 | f
TSPWS_LIB_PATH=$S TSPWS_ENGINE=spectral TSPWS_SPEC_NT=double python tools/random_sweep_spectral.py 70100 50 2>&1 | f
TSPWS_LIB_PATH=$S TSPWS_ENGINE=spectral TSPWS_SPEC_NT=double TSPWS_SPEC_R64=0 python tools/random_sweep_spectral.py 70200 20 2>&1 | f
unset SWEEP_ANYN
python tools/random_sweep_large.py 71000 6 2>&1 | f
} 2>&1 | tee gpurun_out/r6_sweeps2.txt
