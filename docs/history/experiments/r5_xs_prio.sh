#!/bin/bash
# cfg4 side by side: priority of the spectral chain's stream (TSPWS_XS_PRIO) and of the forward stream of the masked call (TSPWS_JK_XFPRIO)
export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
export CFG4_REPS=40
for i in 1 2; do
echo "plain:                  $(TSPWS_XS_PRIO=0 python tools/cfg4_run.py 2>&1 | tail -1)"
echo "chain urgent (default): $(python tools/cfg4_run.py 2>&1 | tail -1)"
echo "chain urgent, fir least: $(TSPWS_XS_PRIO=1 TSPWS_JK_XFPRIO=-1 python tools/cfg4_run.py 2>&1 | tail -1)"
echo "chain least:            $(TSPWS_XS_PRIO=-1 python tools/cfg4_run.py 2>&1 | tail -1)"
done
