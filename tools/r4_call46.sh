#!/bin/bash
O=gpurun_out/r4as; mkdir -p $O
S=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "wall $(( $(date +%s) - S )) s"; tail -2 $O/bench_driver.err
python -c "
import json; d=json.loads([l for l in open('$O/bench_driver.json') if l.startswith('{')][-1]); print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k:(round(v['value']), v.get('roofline',{}).get('traffic')) for k,v in d['configs'].items()})"
