"""Stream sharding across the GPUs of one node.

Audio streams are independent (the reference runs one NeuralModel per stream on its own thread,
NeuralAudio/NeuralModel.h:127), so the multi-GPU path is a pure partition: contiguous ranges of the
arch-sorted stream list, balanced by per-stream cost, one process per GPU; state never moves between
GPUs and there is NO data-path collective.  torch.distributed (RCCL) is only used for the optional
fan-in of outputs / timing barriers.
"""
from typing import List, Sequence, Tuple


def shard_ranges(costs: Sequence[float], world_size: int) -> List[Tuple[int, int]]:
    """Split streams [0, len(costs)) into `world_size` contiguous ranges with near-equal total cost.

    costs[i] is the relative per-sample cost of stream i (e.g. algorithmic bytes or MACs per sample).
    Returns [(begin, end)] per rank; ranges are contiguous, ordered, disjoint and cover everything.
    """
    n = len(costs)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    total = float(sum(costs))
    ranges = []
    begin = 0
    acc = 0.0
    for rank in range(world_size):
        if rank == world_size - 1:
            end = n
        else:
            target = total * (rank + 1) / world_size
            end = begin
            while end < n and acc + costs[end] <= target + 1e-9:
                acc += costs[end]
                end += 1
            # leave at least one stream for each remaining rank when possible
            end = min(end, max(begin, n - (world_size - rank - 1)))
            acc = float(sum(costs[:end]))
        ranges.append((begin, end))
        begin = end
    return ranges


def my_range(costs: Sequence[float], rank: int, world_size: int) -> Tuple[int, int]:
    return shard_ranges(costs, world_size)[rank]
