#!/bin/bash
# usage (GPU box): bash tools/experiments/inv_classes_cfg4.sh variant_abl.so -- k_inv_poly time of cfg4's twelve batched reconstructions per log2(D) class (FL_ABLATE build)
SO=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TSPWS_LIB_PATH=$R/ts-pws_amd/lib/$SO
cd /tmp && export TMPDIR=/tmp
for m in 7f 01 02 04 08 10 20 40; do
  export TSPWS_INV_CLASSES=$m
  rm -rf /tmp/ic_$m
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ic_$m -o b -- python3 $R/tools/cfg4_run.py > /tmp/ic_$m.log 2>&1
  printf "classes %s: " $m
  python3 $R/profiles/summarize_rocpd.py /tmp/ic_$m/b_results.db | grep k_inv_poly
done
