#!/bin/bash
# cfg4 call by call, 150 calls: alone in the process / after calls of another plan; transforms one after the other / side by side with the FIR stream's priority plain, least, most
export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
export NCALLS=150
for other in "" 1; do
echo "== other plan in the process: ${other:-no}"
echo "serial:       $(OTHER_PLANS=$other TSPWS_SPEC_PARALLEL=0 python tools/experiments/r5_cfg4_percall.py)"
echo "fir plain:    $(OTHER_PLANS=$other TSPWS_XS_PRIO=0 python tools/experiments/r5_cfg4_percall.py)"
echo "fir least:    $(OTHER_PLANS=$other TSPWS_XS_PRIO=-1 python tools/experiments/r5_cfg4_percall.py)"
echo "fir most:     $(OTHER_PLANS=$other TSPWS_XS_PRIO=1 python tools/experiments/r5_cfg4_percall.py)"
done
