"""CPU, world_size 2, gloo: the multi-process plumbing bench.py uses for N > 1 (sharding, barrier, max-over-ranks),
with the CPU oracle standing in for the GPU decoder so that 'sharded decode == unsharded decode' is checked too."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fec_testlib as T
from dvbs2rx_amd import shard


def test_shard_ranges_are_group_aligned_and_cover_everything():
    for total, world, G in [(4096, 8, 32), (4096, 3, 32), (100, 4, 32), (32768, 8, 16), (31, 2, 32), (64, 1, 32)]:
        got = [shard.shard_range(total, world, r, G) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        for (a0, a1), (b0, b1) in zip(got, got[1:]):
            assert a1 == b0
        for a0, a1 in got:
            assert a0 % G == 0 and (a1 % G == 0 or a1 == total)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    w, r, _ = shard.init_from_env(backend="gloo")
    assert (w, r) == (world, rank) and dist.get_backend() == "gloo"
    table, G, total, trials = "S2_TABLE_C1", 16, 64, 20
    llr, _ = T.llr_codeword_awgn(table, total, 77, amp=5, sigma=6.3)
    a, b = shard.shard_range(total, world, rank, G)
    shard.barrier_sync()
    out, ret = T.oracle_ldpc_decode(table, llr[a:b], G, trials)
    shard.barrier_sync()
    dt = shard.max_over_ranks(1.0 + rank)           # slowest rank defines the step time
    frames = shard.sum_over_ranks(b - a)
    assert shard.gather_over_ranks(10.0 + rank) == [10.0 + r for r in range(world)]  # per-rank rates, rank order
    q.put((rank, a, b, T.sha(out), ret, dt, frames))
    shard.finalize()


def test_two_ranks_gloo_sharded_decode_equals_unsharded():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    table, G, total, trials = "S2_TABLE_C1", 16, 64, 20
    llr, _ = T.llr_codeword_awgn(table, total, 77, amp=5, sigma=6.3)
    full, fret = T.oracle_ldpc_decode(table, llr, G, trials)
    (r0, a0, b0, h0, ret0, dt0, f0), (r1, a1, b1, h1, ret1, dt1, f1) = res
    assert (a0, b0, a1, b1) == (0, 32, 32, 64)
    assert h0 == T.sha(full[:32]) and h1 == T.sha(full[32:]) and ret0 + ret1 == fret
    assert dt0 == dt1 == 2.0 and f0 == f1 == 64.0


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    shard.init_from_env(backend="gloo")
    a, b = shard.shard_range(32768, world, rank, 32)   # BASELINE config 5: 32768 frames over 8 GPUs, groups of 32
    shard.barrier_sync()
    dt = shard.max_over_ranks(0.5 + 0.01 * rank)
    frames = shard.sum_over_ranks(b - a)
    q.put((rank, a, b, dt, frames))
    shard.finalize()


def test_eight_ranks_gloo_config5_shares():
    """The 8-rank shape of BASELINE config 5 (bench.py --gpus 8 runs the LLR-domain chain on 4096 frames per rank): every rank gets
    exactly 4096 frames = 128 whole groups, the ranges tile the batch, the job's time is the slowest rank's, no data collective."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, (rank, a, b, dt, frames) in enumerate(res):
        assert (rank, a, b) == (r, 4096 * r, 4096 * (r + 1))
        assert abs(dt - 0.57) < 1e-9 and frames == 32768.0
