#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
DVBS2_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --gate first > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc $?"
tail -c 1500 $O/bench_n2.json; tail -5 $O/bench_n2.err
bash tools/r4_lease.sh A
