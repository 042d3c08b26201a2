#!/bin/bash
# tools/prof_ldpc.sh <outdir> -- rocprofv3 kernel-trace stats + two PMC passes of the B4 workload (GPU box).
set -e
OUT=${1:-gpurun_out/prof}
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/exp_tables.py S2_TABLE_B4"
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/$OUT/trace -o t -- $CMD > $REPO/$OUT/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $REPO/$OUT/pmc1 -o p -- $CMD > $REPO/$OUT/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $REPO/$OUT/pmc2 -o p -- $CMD > $REPO/$OUT/pmc2.log 2>&1 || true
cd $REPO
find $OUT -name "*.csv" | head -20
