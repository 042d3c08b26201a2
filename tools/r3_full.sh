#!/bin/bash
# GPU box: the whole -m gpu suite, smoke, default bench
O=gpurun_out/${1:-r3full}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench.log 2>&1
cat $O/pytest_gpu.log $O/smoke.log; tail -c 2500 $O/bench.log
