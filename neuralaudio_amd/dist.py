"""One-process-per-GPU plumbing over torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

The data path needs no collective (streams are independent, see sharding.py).  These helpers cover what a
multi-GPU host does around it: rendezvous, the barrier + max-over-ranks used for timing, and the optional
fan-in (all_gather) of per-rank output shards when one consumer wants every stream's output.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend="nccl", device=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kwargs = {}
    if device is not None and backend == "nccl":
        kwargs["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return True


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of one python float (the per-rank elapsed time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_shards(local, ranges):
    """Fan-in: every rank contributes its [streams_r, n] output shard, everybody gets [total_streams, n].

    Shards may have different sizes (cost-balanced ranges); they are padded to the largest for all_gather.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    sizes = [b - a for a, b in ranges]
    biggest = max(sizes)
    n = local.shape[1]
    padded = torch.zeros(biggest, n, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
