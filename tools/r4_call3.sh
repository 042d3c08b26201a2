#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "host or async or chunk" 2>&1 | tail -3
python tools/host_entry_ab.py 2>/dev/null >> $O/host_ab.log
for plan in "512" "512,3584" "512,1792" "512,1536,1536,512" "512,1024" "256,256,1024" "512,3072,512" "1024"; do
  DVBS2_HOST_PLAN=$plan python tools/host_entry_ab.py 4096 2>/dev/null | sed "s/^default/plan=$plan/" >> $O/host_ab.log
done
python tools/host_entry_ab.py 2>/dev/null >> $O/host_ab.log
cat $O/host_ab.log
