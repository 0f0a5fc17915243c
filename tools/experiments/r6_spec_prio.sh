#!/bin/bash
# round 6: wave priority of the spectral chain's transform passes (s_setprio 3) beside the FIR kernels: cfg1, cfg2, cfg4, back to back on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
bash tools/variants_cfg.sh "cfg1 40" libtspws_hip.so variant_prio3.so
bash tools/variants_cfg.sh "cfg2 40" libtspws_hip.so variant_prio3.so
bash tools/variants_cfg.sh "c:500:20000 40" libtspws_hip.so variant_prio3.so
for so in libtspws_hip.so variant_prio3.so libtspws_hip.so variant_prio3.so; do printf "%-24s" $so; TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/$so CFG4_REPS=30 python tools/cfg4_run.py 2>&1 | grep -v amdgpu; done
} 2>&1 | tee gpurun_out/r6_spec_prio.txt
