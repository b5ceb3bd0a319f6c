"""Multi-GPU use: envs are independent, so a population is sharded by contiguous global env
ids — rank r owns ``[r * envs_per_rank, (r + 1) * envs_per_rank)`` — with no data-path
collective.  The only exchange is a sum of the 8-entry int64 metrics vector
(``torch.distributed`` all-reduce; backend ``nccl`` = RCCL on ROCm, ``gloo`` on CPU)."""
import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(total_envs, rank=None, world=None):
    """(env_id_base, count) of this rank for a population of ``total_envs`` envs; the remainder
    is spread over the first ranks, every env id belongs to exactly one rank."""
    if rank is None or world is None:
        rank, world = rank_world()
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    q, r = divmod(int(total_envs), world)
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


def init_process_group(backend=None):
    """torch.distributed bootstrap from the torchrun environment (MASTER_ADDR defaults to
    127.0.0.1).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def allreduce_metrics(metrics, device=None):
    """Sum an int64 metrics vector over all ranks; returns a numpy array (identity when not
    distributed)."""
    import torch
    import torch.distributed as dist
    m = np.asarray(metrics, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return m.copy()
    t = torch.from_numpy(m.copy())
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
