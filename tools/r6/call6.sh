#!/bin/bash
O=gpurun_out/r6f; mkdir -p $O
python tools/r6/exp_prio.py S2_TABLE_B4 2.0 50 4096 > $O/prio_b4.txt 2>&1; cat $O/prio_b4.txt
python tools/r6/exp_prio.py S2_TABLE_C1 0.5 25 16384 > $O/prio_c1.txt 2>&1; cat $O/prio_c1.txt
python tools/r6/exp_prio.py S2_TABLE_B7 5.3 50 4096 > $O/prio_b7.txt 2>&1; cat $O/prio_b7.txt
