#!/bin/bash
# cfg2: the spectral chain's stream beside k_fwd_tl with the highest / lowest stream priority (40 timed calls each)
export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
for i in 1 2 3; do for pr in 0 1 -1; do
echo "side prio $pr: $(TSPWS_SIDE_PRIO=$pr python tools/cfg_bench.py cfg2 40 2>&1 | tail -1)"
done; done
