"""Stream sharding across the GPUs of one node.

Audio streams are independent (the reference runs one NeuralModel per stream on its own thread,
NeuralAudio/NeuralModel.h:127), so the multi-GPU path is a pure partition: contiguous ranges of the
arch-sorted stream list, balanced by per-stream cost, one process per GPU; state never moves between
GPUs and there is NO data-path collective.  torch.distributed (RCCL) is only used for the optional
fan-in of outputs / timing barriers.
"""
import ctypes as C
from typing import List, Sequence, Tuple


def shard_ranges(costs: Sequence[float], world_size: int) -> List[Tuple[int, int]]:
    """Split streams [0, len(costs)) into `world_size` contiguous ranges with near-equal total cost.

    costs[i] is the relative per-sample cost of stream i (e.g. NA_ModelStreamCost of its model).
    Returns [(begin, end)] per rank; ranges are contiguous, ordered, disjoint and cover everything.
    This is the C++ host's partition (libNeuralAudioCAPI: NA_ShardByCost, csrc/multi_gpu.cpp) -- the one-process-per-GPU hosts
    (bench.py) and the single-process multi-GPU host (NA_Multi*) shard with the same code.
    """
    from . import capi
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    lib = capi.load_library()
    n = len(costs)
    arr = (C.c_double * max(n, 1))(*[float(c) for c in costs])
    bounds = (C.c_int * (world_size + 1))()
    if lib.NA_ShardByCost(arr, n, world_size, bounds) != 0:
        raise RuntimeError(capi.last_error())
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world_size)]


def my_range(costs: Sequence[float], rank: int, world_size: int) -> Tuple[int, int]:
    return shard_ranges(costs, world_size)[rank]
