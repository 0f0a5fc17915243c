#!/bin/bash
# cfg2: the spectral chain's stream beside k_fwd_tl with the highest / lowest stream priority, and the trace-lane workgroup length with it
export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
for pr in 0 1 -1; do for st in 12 24; do
echo "side prio $pr tl steps $st: $(TSPWS_SIDE_PRIO=$pr TSPWS_SPEC_TLSTEPS=$st python tools/cfg2_run.py 2>&1 | tail -1)"
done; done
