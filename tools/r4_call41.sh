#!/bin/bash
O=gpurun_out/r4an; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "two_handles or group_stop or kernel_variant or async or enqueue" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
for a in "4096 2.0 6 2 0" "2048 2.0 8 2 0" "8192 2.0 4 2 0" "4096 2.0 6 3 0"; do timeout 600 python tools/exp_awgn_pipe.py $a 2>/dev/null | tail -1 >> $O/pipe.txt; done; cat $O/pipe.txt
