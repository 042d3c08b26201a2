#!/bin/bash
# GPU box, round 3 run 2: group-synchronous stop (parity + operating point), low-register builds A/B
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "not every_table_bit_exact" 2>&1 | tail -5 > $O/pytest_gs.log
DVBS2_GROUP_SYNC=0 timeout 600 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "near_threshold or group_of_one or async or multi_chunk or parity_in_records" 2>&1 | tail -3 > $O/pytest_nogs.log
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_bch_demap_gpu.py -x -q -m gpu -k "every_table_bit_exact and (policy or classic) or chain" 2>&1 | tail -3 > $O/pytest_tables.log
timeout 300 python bench.py --only config2_awgn --no-cpu-baseline > $O/bench_awgn.log 2>&1
T="S2_TABLE_B8:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B10:50:4096 S2X_TABLE_B19:50:4096 S2X_TABLE_B20:50:4096 S2X_TABLE_B21:50:4096 S2X_TABLE_B24:50:4096 S2_TABLE_C9:25:16384 S2_TABLE_C10:25:16384"
echo "== default" > $O/lr_ab.log; timeout 600 python tools/exp_tables.py $T 2>&1 | grep -v amdgpu.ids >> $O/lr_ab.log
echo "== lr" >> $O/lr_ab.log; DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_lr.so timeout 600 python tools/exp_tables.py $T 2>&1 | grep -v amdgpu.ids >> $O/lr_ab.log
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_lr.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "every_table_bit_exact and (B8 or B9 or B10 or B11 or B19 or B2 or C9 or C10)" 2>&1 | tail -3 > $O/pytest_lr.log
tail -3 $O/pytest_gs.log $O/pytest_nogs.log $O/pytest_tables.log $O/pytest_lr.log; cat $O/lr_ab.log; tail -c 1200 $O/bench_awgn.log
