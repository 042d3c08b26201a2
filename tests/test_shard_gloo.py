"""CPU, world_size 2, gloo: the multi-process plumbing bench.py uses for N > 1 (sharding, barrier, max-over-ranks),
with the CPU oracle standing in for the GPU decoder so that 'sharded decode == unsharded decode' is checked too."""
import os
import socket
import subprocess
import sys

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


@pytest.mark.gpu
def test_bench_n2_branch_runs_on_one_gpu():
    """the N > 1 branch of bench.py executed once before a driver points an 8-GPU node at it -- two ranks under
    torch.distributed.run exactly as the driver launches them (rendezvous on 127.0.0.1), process group on gloo so that both ranks may
    share this box's one GPU (DVBS2_DIST_BACKEND; RCCL refuses two ranks on one device). Checks the line the driver parses: n_gpus,
    the whole-job value, per_rank (gathered over the ranks), config 5 on every rank, and the host feed of every rank."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DVBS2_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gate", "first",
           "--frames", "1024"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["config"]["frames_per_gpu"] == 1024 and "bit-exact" in d["parity"]
    pr = d["per_rank"]
    assert len(pr["frames_per_s"]) == 2 and all(v > 0 for v in pr["frames_per_s"])
    # whole-job value = frames of both ranks / the slowest rank's time: between one and two times the slower rank's own rate
    assert 0.9 * pr["min"] <= d["value"] <= 2.0 * pr["max"] * 1.05
    c5 = d["configs"]["config5"]
    assert c5["frames_total"] == 2 * c5["frames_per_gpu"] and c5["value"] > 0 and c5["roofline"]["kernel"].startswith("ldpc_layered_kernel<32")
    host = d["configs"]["config2_host"]["ranks"]
    for mode in ("pageable", "page_locked"):
        assert len(host[mode]["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in host[mode]["per_rank_frames_per_s"])
    assert d["fallback_rounds"] == 0


@pytest.mark.gpu
def test_rccl_process_group_at_world_size_one():
    """The RCCL side of the N > 1 launch on the one GPU of the test box: DVBS2_FORCE_PG=1 makes shard.init_from_env
    initialise the `nccl` (= RCCL) process group with device_id at WORLD_SIZE = 1; barrier_sync, max / sum / gather over ranks then run
    their all-reduces on DEVICE tensors through RCCL. In a process of its own (a process group is per process)."""
    code = r"""
import os, sys
sys.path.insert(0, os.path.join(%r, "gr-dvbs2rx_amd", "python"))
import torch, torch.distributed as dist
from dvbs2rx_amd import shard
world, rank, local = shard.init_from_env()
assert (world, rank, local) == (1, 0, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
torch.cuda.set_device(local)
shard.barrier_sync()
assert shard.max_over_ranks(3.25, device=dev) == 3.25
assert shard.sum_over_ranks(2.5, device=dev) == 2.5
assert shard.gather_over_ranks(7.0, device=dev) == [7.0]
t = torch.arange(1024, dtype=torch.float32, device=dev)
dist.all_reduce(t)
assert float(t.sum().item()) == float(sum(range(1024)))
shard.finalize()
print("rccl ok")
""" % T.ROOT
    env = dict(os.environ, DVBS2_FORCE_PG="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DVBS2_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr
