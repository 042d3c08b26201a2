#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "kernel_variant_policy or (every_table_bit_exact and B9)" 2>&1 | tail -3 > $O/test.txt; cat $O/test.txt
DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_TIMING_WAVES=1 python tools/exp_tables.py S2_TABLE_B4:50:512 > $O/timing_b4.txt 2>&1
grep -v "cycles/sweep" $O/timing_b4.txt | tail -12; grep "cycles/sweep" $O/timing_b4.txt | tail -100 | awk '{print $2,$4,$6,$8,$9}' | tr '\n' ';'
