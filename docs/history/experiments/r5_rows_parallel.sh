#!/bin/bash
# cfg4: the rows-in-columns spectral chain beside k_fwd_lds (the default; TSPWS_SPEC_PARALLEL=0 on the sweeps build: one after the other) with the end-of-round kernels (transposition in 256-sample
# workgroups without atomics, fold with a 4-KB tile per wave) -- mid-round the pair took 1.70 ms side by side against 1.05 one after the other
export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
export CFG4_REPS=40
for i in 1 2 3; do
echo "serial:   $(TSPWS_SPEC_PARALLEL=0 python tools/cfg4_run.py 2>&1 | tail -1)"
echo "parallel: $(python tools/cfg4_run.py 2>&1 | tail -1)"
done
