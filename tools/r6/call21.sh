#!/bin/bash
# GPU box, round 6 call 21: sanity after the launch-flag cleanup; short tables with / without the packed nodes at an operating point
O=gpurun_out/r6x; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "policy or (test_every_table_bit_exact and (pr-packed or pr-plain or pr-byte) and (C2 or C3 or C4 or B4 or C10))" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for r in 1 2; do for v in 1 0; do
  DVBS2_PR_V2=$v python tools/exp_awgn2.py 16384 1.0 7 S2_TABLE_C3 25 2>&1 | tail -1 | sed "s/^/[PR_V2=$v] /"
  DVBS2_PR_V2=$v python tools/exp_awgn2.py 16384 2.2 7 S2_TABLE_C4 25 2>&1 | tail -1 | sed "s/^/[PR_V2=$v] /"
done; done > $O/awgn.txt 2>&1; cat $O/awgn.txt
