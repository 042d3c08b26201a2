#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_fw28.so python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and (plain or policy or classic)" 2>&1 | tail -2
bash tools/ab3.sh $O/ab3.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_fw28.so" S2_TABLE_B8:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_C7:25:16384 S2_TABLE_C8:25:16384 S2_TABLE_B4:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B11:50:4096 > $O/ab3_res.log 2>&1
cat $O/ab3_res.log
for t in S2_TABLE_B5 T2_TABLE_A3; do for lib in libdvbs2_fec_hip.so libdvbs2_fec_hip_fw28.so; do echo -n "$t plain+solo $lib: "; DVBS2_V2=0 DVBS2_CHAIN_V2=0 DVBS2_SOLO=1 DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/$lib python tools/exp_tables.py $t:50:4096 2>/dev/null | awk '{print $8, $9}'; done; done
