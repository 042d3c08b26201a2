#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
for rep in 1 2; do
for cs in 0 1; do
  DVBS2_HOST_COPY_STREAM=$cs python tools/host_entry_ab.py 2>/dev/null >> $O/host_ab.log
done
python tools/host_entry_ab.py 2>/dev/null >> $O/host_ab.log
done
for ch in 256 1024; do DVBS2_HOST_CHUNK=$ch python tools/host_entry_ab.py 4096 2>/dev/null >> $O/host_ab.log; done
cat $O/host_ab.log
