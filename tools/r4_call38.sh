#!/bin/bash
O=$PWD/gpurun_out/r4ak; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQC_|INST_CACHE|SQ_INST_LEVEL|SQ_WAIT_INST" | head -60 > $O/counters.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --gate none --only config3,config4,config5,config5_s2x"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU -d $O/pmc_ic -o p -- $CMD > $O/pmc_ic.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_ic/p_results.db > $O/ic_summary.txt 2>&1
rm -rf $O/pmc_ic
head -40 $O/counters.txt; grep "ldpc_layered" $O/ic_summary.txt | cut -c1-400
