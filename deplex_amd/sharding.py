"""Frame sharding across GPUs.  Frames are independent (an extractor holds only per-call scratch, reference
plane_extractor.cpp:281,428), so a batch shards into contiguous frame ranges, one per rank, with no data-path
collective."""
from __future__ import annotations


def shard_range(n_frames: int, rank: int, world: int) -> tuple:
    """frames[g*B/G : (g+1)*B/G] for rank g of G; covers 0..n_frames exactly once, sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    return (n_frames * rank) // world, (n_frames * (rank + 1)) // world
