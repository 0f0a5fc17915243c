#!/bin/bash
# round-4 small experiments (GPU box): side-stream priority of the coarse-scale kernel (cfg2, cfg3 finish), phase normalisation
# inside the fused forward kernel (timing ablation build variant_nonorm.so: results wrong, only the clock counts)
cd $GRAFT_REPO_ROOT
for pr in 0 1 -1; do echo "side prio $pr"; for i in 1 2; do TSPWS_SIDE_PRIO=$pr python tools/cfg2_run.py | tail -1; done; TSPWS_SIDE_PRIO=$pr python tools/fwd_bench.py | tail -1; done
echo "normalisation ablation"
bash tools/variants.sh "" libtspws_hip.so variant_nonorm.so
FWD_MEXHAT=1 bash tools/variants.sh "131072 100" libtspws_hip.so variant_nonorm.so
