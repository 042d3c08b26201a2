#!/bin/bash
# tools/profile_round.sh <tag> -- GPU box: rocprofv3 kernel-trace stats of bench.py and separate PMC passes
# (SQ activity; FETCH_SIZE; WRITE_SIZE -- TCC counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Summaries (text) go to gpurun_out/<tag>/ ; copy the ones to keep into profiles/.
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --gate none --only config3,config4,config5,config5_s2x"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o p -- $CMD > $OUT/pmc_l2.log 2>&1
cd $REPO
{
  echo "# rocprofv3 summary ($TAG): $CMD"
  echo "# bench line:"; grep '^{' $OUT/trace.log | tail -1
  python tools/pmc_summary.py $OUT/trace/t_results.db $OUT/pmc_sq/p_results.db $OUT/pmc_fetch/p_results.db $OUT/pmc_write/p_results.db $OUT/pmc_l2/p_results.db
} > $OUT/summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_l2
cat $OUT/summary.txt
