#!/bin/bash
# usage: bash profiles/collect_pmc_icache.sh TAG [ENV=VAL...]  -- instruction-fetch counters for the forward kernels
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmci_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_LEVEL\|WAIT_INST\)[A-Z_]*" | sort -u > $OUT/avail.txt
run() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra > $OUT/$n.log 2>&1; }
run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run c SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES
cd $R
python3 profiles/summarize_pmc.py $OUT > gpurun_out/pmci_$TAG.txt 2>&1
cat $OUT/avail.txt | tr "\n" " "; echo
grep -A 16 "^k_fwd" gpurun_out/pmci_$TAG.txt
