#!/bin/bash
mkdir -p gpurun_out
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1 TSPWS_SPEC_NSMAX=2048 TSPWS_SPEC_NSW=16 TSPWS_SPEC_NTB=2
PMC_CMD="tools/cfg2_run.py" PMC_SETS="sq sq2 fetch write" bash profiles/collect_pmc.sh r05fold > /dev/null 2>&1
grep -A22 "k_spec_fold\|k_spec_inv\|k_spec_mid<false>\|k_spec_fwd_last" gpurun_out/pmc_r05fold.txt | head -150
