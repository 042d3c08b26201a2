#!/bin/bash
O=gpurun_out/r4ay; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['parity'][:40], {k:round(v['value']) for k,v in d['configs'].items()})"
