#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_ldpc.log
timeout 300 python tools/exp_tables.py S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C10:25:16384 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 2>&1 | grep -v amdgpu.ids > $O/tables.log
timeout 300 python bench.py --only config2_awgn --no-cpu-baseline --gate first > $O/bench.log 2>&1
cat $O/pytest_ldpc.log $O/tables.log; tail -c 600 $O/bench.log
