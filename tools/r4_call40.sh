#!/bin/bash
O=gpurun_out/r4am; mkdir -p $O
timeout 2400 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "kernel_variant_policy" 2>&1 | tail -3 > $O/test.txt; cat $O/test.txt
for rep in 1 2; do for e in "X=1" "DVBS2_SOFT_BARRIER=0"; do echo "== $e" >> $O/b9.txt; env $e python tools/exp_tables.py S2_TABLE_B9:50:4096 2>&1 | grep fr/s | cut -c1-90 >> $O/b9.txt; done; done; cat $O/b9.txt
