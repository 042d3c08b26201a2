#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "two_handles_pipelined or group_stop" 2>&1 | tail -3 > $O/test.txt; cat $O/test.txt
timeout 600 python bench.py --no-cpu-baseline --only config2_awgn --gate first > $O/bench_awgn.json 2> $O/bench_awgn.err; tail -2 $O/bench_awgn.err
python -c "
import json; d=json.loads([l for l in open('$O/bench_awgn.json') if l.startswith('{')][-1]); a=d['configs']['config2_awgn']; print(round(d['value']), round(a['value']), a['frac_of_proportional_rate'], a['pipelined'])"
timeout 900 python tools/soft_sweep.py > $O/soft_sweep.txt 2>&1; cat $O/soft_sweep.txt
