#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact or near_threshold or never or full_batch" 2>&1 | tail -2
bash tools/ab_tables.sh $O/ab.log libdvbs2_fec_hip_fw0.so libdvbs2_fec_hip.so S2_TABLE_B4:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2X_TABLE_B3:50:4096 S2X_TABLE_B11:50:4096 S2_TABLE_B4:50:4096 > $O/ab_res.log 2>&1
cat $O/ab_res.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4l/bench_driver_cmd.json').readline())
print('driver-cmd: headline',round(d['value']), {k:round(c['value']) for k,c in d['configs'].items()}, 'copy', round(d['device_copy']['read_plus_write_gbs']))
PY
