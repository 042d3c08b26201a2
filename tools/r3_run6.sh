#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_ldpc.log
timeout 1200 python tools/all_tables_perf.py > $O/all_tables.md 2>$O/all_tables.err
cat $O/pytest_ldpc.log; cat $O/all_tables.md
