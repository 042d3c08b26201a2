#!/bin/bash
# GPU box, round 6 call 12: B1 / S2X B1 through the parity-in-records kernel (policy by rule) -- bit-exactness; policy sweep of every table (packed x one-frame)
O=gpurun_out/r6l; mkdir -p $O
timeout 600 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 -k "test_every_table_bit_exact and (S2_TABLE_B1- or S2X_TABLE_B1-)" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 2400 python tools/policy_sweep.py > $O/sweep.txt 2> $O/sweep.err; tail -3 $O/sweep.txt
