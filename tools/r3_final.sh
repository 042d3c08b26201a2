#!/bin/bash
# GPU box, end of round 3: whole -m gpu suite, smoke, default bench, rocprofv3 passes, every table
bash tools/r3_full.sh r3final > /dev/null 2>&1
bash tools/profile_round.sh r03 > /dev/null 2>&1
timeout 1500 python tools/all_tables_perf.py > gpurun_out/r3final/all_tables.md 2> gpurun_out/r3final/all_tables.err
cat gpurun_out/r3final/pytest_gpu.log gpurun_out/r3final/smoke.log | tail -4; tail -c 400 gpurun_out/r3final/bench.log; tail -3 gpurun_out/r3final/all_tables.md
