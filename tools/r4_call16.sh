#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact" 2>&1 | tail -2
bash tools/ab3.sh $O/ab3.log "libdvbs2_fec_hip_r3.so libdvbs2_fec_hip.so" S2_TABLE_C5:25:16384 S2_TABLE_C6:25:16384 > $O/ab3_res.log 2>&1
cat $O/ab3_res.log
python tools/policy_sweep.py S2_TABLE_B5 S2_TABLE_B6 T2_TABLE_A3 S2X_TABLE_C5 S2X_TABLE_C6 S2X_TABLE_C4 T2_TABLE_B3 S2X_TABLE_B22 2>&1 | tee $O/policy12.log
