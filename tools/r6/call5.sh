#!/bin/bash
# GPU box, round 6 call 5: itemise the operating-point losses (configs 2, 3's LDPC table, 4)
O=gpurun_out/r6e; mkdir -p $O
python tools/r6/opp_itemise.py S2_TABLE_B4 2.0 50 4096 512 > $O/opp_b4.txt 2>&1; cat $O/opp_b4.txt
python tools/r6/opp_itemise.py S2_TABLE_C1 0.5 25 16384 1024 > $O/opp_c1.txt 2>&1; cat $O/opp_c1.txt
python tools/r6/opp_itemise.py S2_TABLE_B7 5.3 50 4096 512 > $O/opp_b7.txt 2>&1; cat $O/opp_b7.txt
