#!/bin/bash
# usage (GPU box, repo root): bash profiles/collect_pmc.sh TAG            (PMC_CMD="tools/cfg2_run.py" PMC_SETS="sq sq2": another script / fewer passes)
# Separate rocprofv3 --pmc passes (counters never combined with sys/hip traces), csv output into gpurun_out/pmc_TAG/
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/${PMC_CMD:-bench.py --steps 2 --warmup 1 --no-cpu} > $OUT/$n.log 2>&1
}
want() { [ -z "$PMC_SETS" ] || [[ " $PMC_SETS " == *" $1 "* ]]; }
want sq && run sq   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
want sq2 && run sq2  SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
want mfma && run mfma SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU
want tcc && run tcc  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
want fetch && run fetch FETCH_SIZE
want write && run write WRITE_SIZE
want tcp && run tcp  TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
cd $R
python3 profiles/summarize_pmc.py $OUT > gpurun_out/pmc_$TAG.txt 2>&1
cat gpurun_out/pmc_$TAG.txt
