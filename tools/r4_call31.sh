#!/bin/bash
O=gpurun_out/r4ad; mkdir -p $O
timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact or near_threshold or never or full_batch or parity_in_records" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip_base.so libdvbs2_fec_hip.so" S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C2:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C8:25:16384 S2X_TABLE_C9:25:16384 S2X_TABLE_C10:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
