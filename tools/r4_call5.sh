#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "near_threshold or group_stop or full_batch or syndrome or never" 2>&1 | tail -3
for nf in 4096 512 8192; do python tools/exp_awgn2.py $nf 2.0 5 2>/dev/null >> $O/awgn.log; done
python tools/exp_awgn2.py 4096 2.0 5 2>/dev/null >> $O/awgn.log
cat $O/awgn.log
