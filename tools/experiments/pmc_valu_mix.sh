#!/bin/bash
# usage (GPU box): bash tools/experiments/pmc_valu_mix.sh TAG script.py -- what the VALU instructions of a run are (FP64 FMA / MUL / ADD / transcendental, integer, moves)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/pmcv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*" | sort -u > $R/gpurun_out/pmcv_${TAG}_avail.txt
run() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/$CMD > $OUT/$n.log 2>&1; }
CMD="$*"
run f64a SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64
run f64b SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
run cyc SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU
cd $R
python3 profiles/summarize_pmc.py $OUT > gpurun_out/pmcv_$TAG.txt 2>&1
tail -5 $OUT/*.log | head -40
cat gpurun_out/pmcv_${TAG}_avail.txt | tr '\n' ' '; echo; grep -A22 "k_fwd_lds\|k_inv_poly\|k_fwd_tl" gpurun_out/pmcv_$TAG.txt | head -150
rm -rf $OUT
