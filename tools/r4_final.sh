#!/bin/bash
# GPU box, round 4 checkpoint: whole -m gpu suite, smoke, default bench, rocprofv3 passes (tag as $1), every table
TAG=${1:-r04}
bash tools/r3_full.sh ${TAG}full > /dev/null 2>&1
bash tools/profile_round.sh $TAG > /dev/null 2>&1
timeout 1500 python tools/all_tables_perf.py > gpurun_out/${TAG}full/all_tables.md 2> gpurun_out/${TAG}full/all_tables.err
cat gpurun_out/${TAG}full/pytest_gpu.log gpurun_out/${TAG}full/smoke.log | tail -4; tail -c 400 gpurun_out/${TAG}full/bench.log; tail -3 gpurun_out/${TAG}full/all_tables.md
