#!/bin/bash
# round 4 (second session): VALU rates incl. fp32 / packed fp32, any-order launch probe, pipelined operating point
O=gpurun_out/r4t; mkdir -p $O
tools/bin/valu_rate > $O/valu_rate.txt 2>&1
tools/bin/anyorder > $O/anyorder.txt 2>&1
for a in "4096 2.0 6 2 0" "4096 2.0 6 2 1" "4096 2.0 6 3 0" "2048 2.0 8 2 0" "8192 2.0 4 2 0"; do
  timeout 600 python tools/exp_awgn_pipe.py $a >> $O/pipe.txt 2>&1
done
cat $O/anyorder.txt $O/pipe.txt; cat $O/valu_rate.txt
