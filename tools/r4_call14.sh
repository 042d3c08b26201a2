#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/ab_tables.sh $O/ab.log libdvbs2_fec_hip_fw0.so libdvbs2_fec_hip.so S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C1:25:16384 S2_TABLE_B2:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B6:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B10:50:4096 S2X_TABLE_B10:50:4096 S2_TABLE_C5:25:16384 S2_TABLE_C7:25:16384 S2_TABLE_C9:25:16384 S2_TABLE_C10:25:16384 > $O/ab_res.log 2>&1
cat $O/ab_res.log
for nf in 4096; do python tools/exp_awgn2.py $nf 2.0 5 2>/dev/null; done
