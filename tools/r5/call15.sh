#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
python tools/exp_chain_awgn.py 2>&1 | tee $O/chain_awgn.txt | tail -4
