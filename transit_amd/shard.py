"""Wavenumber-axis sharding: every stage after the line
sweep is independent per wavenumber and each line touches a contiguous window
of bins, so rank r of N owns the coarse bins [lo, hi) and nothing else."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_bounds(nwn: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split of nwn bins over world ranks."""
    if not (0 <= rank < world) or world > nwn:
        raise ValueError("bad shard request: nwn=%d world=%d rank=%d" % (nwn, world, rank))
    base, rem = divmod(nwn, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_bounds(nwn: int, world: int) -> List[Tuple[int, int]]:
    return [shard_bounds(nwn, world, r) for r in range(world)]


def stitch(parts: Sequence[np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(p) for p in parts])
