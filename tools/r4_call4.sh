#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
for nf in 4096 512 1024 2048 8192; do python tools/exp_awgn2.py $nf 2.0 5 2>/dev/null >> $O/awgn.log; done
DVBS2_TIMING=1 python tools/exp_awgn2.py 512 2.0 2 > $O/awgn_timing.log 2>&1
DVBS2_GROUP_SYNC=0 python tools/exp_awgn2.py 4096 2.0 5 2>/dev/null | sed 's/^/gsync=0 /' >> $O/awgn.log
cat $O/awgn.log; grep -a "timing" $O/awgn_timing.log | tail -4
