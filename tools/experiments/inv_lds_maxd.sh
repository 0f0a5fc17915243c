#!/bin/bash
# round 4 (GPU box): LDS-staged inverse for the octaves up to decimation TSPWS_INV_LDS_MAXD (0 = per-lane form everywhere)
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "forward_inverse or golden or jackknife" 2>&1 | grep -v "Warning: Fold\|waveletFamily" | tail -2)
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  export TSPWS_INV_LDS_MAXD=$m
  rm -rf /tmp/il_$m
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/il_$m -o b -- python3 $GRAFT_REPO_ROOT/tools/cfg4_run.py > /tmp/il_$m.log 2>&1
  printf "maxd %s: " $m; tail -1 /tmp/il_$m.log
  python3 $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/il_$m/b_results.db | grep k_inv_poly
  python3 $GRAFT_REPO_ROOT/tools/fwd_bench.py | tail -1
done
