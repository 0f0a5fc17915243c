#!/bin/bash
# cfg2 whole call (spectral chain beside k_fwd_tl) with three forms of k_spec_fold_mfma: f2 = 64-byte piece stores, no LDS, one operand set ahead (120 VGPRs);
# f1 = + rows through the LDS tile (126 VGPRs, 16 KB of LDS); shipped = + three operand sets (156 VGPRs)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do for v in variant_f2 variant_f1 libtspws_hip; do
  echo "$v: $(TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$v.so python tools/cfg_bench.py cfg2 40 | tail -1)"
done; done
