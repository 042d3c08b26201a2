#!/bin/bash
# GPU box, round 6 call 19: packed nodes in the parity-in-records kernel, layers of degree >= 5 only -- bit-exactness incl. fuzz with the build forced
O=gpurun_out/r6v; mkdir -p $O
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "test_every_table_bit_exact and pr-packed" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
DVBS2_PR_V2=1 timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "(C2 or C3 or C4 or group or near or saturation or counters) and not every_table" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -2 $O/pytest2.log
DVBS2_PR=1 DVBS2_PR_V2=1 timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "(B4 or near or saturation or counters or full_batch) and not every_table" > $O/pytest3.log 2>&1; echo "pytest3 rc $?"; tail -2 $O/pytest3.log
(DVBS2_PR=1 DVBS2_PR_V2=1 python tools/fuzz_ldpc.py 150 61 2>&1 | tail -2) > $O/fuzz_prv2_all.log &
(DVBS2_PR_V2=1 python tools/fuzz_ldpc.py 150 62 2>&1 | tail -2) > $O/fuzz_prv2.log &
wait; tail -1 $O/fuzz_prv2_all.log $O/fuzz_prv2.log
