#!/bin/bash
# GPU box, round 6 call 4: host-pointer chain entry + BCH frame ranges
O=gpurun_out/r6d; mkdir -p $O
timeout 1200 python -m pytest tests/test_bch_demap_gpu.py tests/test_host_blocks.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
