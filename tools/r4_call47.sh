#!/bin/bash
O=gpurun_out/r4at; mkdir -p $O
DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_TIMING_WAVES=1 python tools/exp_tables.py S2_TABLE_B4:50:512 > $O/timing_b4_packed.txt 2>&1
grep -v "cycles/sweep" $O/timing_b4_packed.txt | tail -9; grep "hazard phases" $O/timing_b4_packed.txt | tail -2; grep "layer " $O/timing_b4_packed.txt | tail -90 | awk '{print $2,$4,$10}' | tr '\n' ';'
