"""Frame sharding across the GPUs of one node: FECFRAME groups are independent, so the batch is
split into contiguous, G-aligned ranges -- one process per GPU, NO data-path collective. torch.distributed is
only used for the launch rendezvous, the timing barrier and the max-over-ranks reduction of the elapsed time."""
import os

import torch
import torch.distributed as dist


def shard_range(total_frames, world, rank, group_size):
    """Contiguous [start, stop) of rank's frames; every boundary is a multiple of group_size so that the
    reference's batch grouping (frames [G*g, G*g+G) share an iteration count) is preserved."""
    n_groups = (total_frames + group_size - 1) // group_size
    g0 = (n_groups * rank) // world
    g1 = (n_groups * (rank + 1)) // world
    return min(g0 * group_size, total_frames), min(g1 * group_size, total_frames)


def init_from_env(backend=None):
    """Returns (world, rank, local_rank). Initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DVBS2_FORCE_PG=1: initialise the process group at WORLD_SIZE = 1 as well (tests: the RCCL rendezvous and device all-reduce of the
    # N > 1 launch, exercised on a one-GPU box; needs MASTER_ADDR / MASTER_PORT like any launch)
    force = os.environ.get("DVBS2_FORCE_PG", "") not in ("", "0")
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("DVBS2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            # one rank per GPU is the deployment; more ranks than GPUs (a one-GPU test box) wrap around here, BEFORE the
            # device is handed to the process group, and RCCL then refuses two ranks on one device unless the caller
            # chose DVBS2_DIST_BACKEND=gloo for that experiment
            ndev = max(1, torch.cuda.device_count())
            kw["device_id"] = torch.device("cuda", local % ndev)
        if world == 1:
            kw.update(rank=0, world_size=1)
        dist.init_process_group(backend=backend, **kw)
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())
    return world, rank, local


def barrier_sync():
    """Device work done on this rank, then every rank arrived, then the barrier's own device work done."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _reduce_device(device):
    # gloo (CPU tests; or DVBS2_DIST_BACKEND=gloo to run several ranks on one GPU) reduces host tensors
    return None if dist.is_initialized() and dist.get_backend() == "gloo" else device


def max_over_ranks(value, device=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """One float per rank, in rank order, on every rank (bench.py: per-rank rates and their spread)."""
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=_reduce_device(device))
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu().tolist()]


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
