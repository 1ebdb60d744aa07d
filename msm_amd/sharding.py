"""Stream sharding over GPUs.

The reference evolves its streams one after the other and independently
(simulator/src/main.rs:43-79), so a multi-GPU run is a block partition of the
seed list with no data-path exchange.  The only cross-GPU product is the
ensemble density sum_s |psi_s|^2 (the synthesizer's psi2 accumulation,
synthesizer/src/lib.rs:217-229), summed over ranks with one all-reduce.
"""
from typing import List, Tuple


def partition_streams(n_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """(first, count) of the block of streams owned by `rank`; blocks are ceil(n/world) long,
    trailing ranks may own fewer or none."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    per = (n_streams + world - 1) // world
    first = min(rank * per, n_streams)
    return first, max(0, min(per, n_streams - first))


def local_seeds(seeds: List[int], world: int, rank: int) -> List[int]:
    first, count = partition_streams(len(seeds), world, rank)
    return seeds[first:first + count]


def allreduce_density(dist, dens, async_op: bool = False):
    """Sum the per-rank sum_s |psi_s|^2 grids in place (RCCL on GPUs, gloo in CPU tests).
    Divide by the global stream count for the ensemble mean."""
    return dist.all_reduce(dens, op=dist.ReduceOp.SUM, async_op=async_op)
