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


def balanced_bounds(cost: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """Contiguous split of the bins into `world` shards of near-equal COST (SURVEY section 8e:
    "balance slices by the work, not by N_wn").  cost[j] >= 0 is the work bin j causes; the
    k-th cut is put where the running sum crosses k/world of the total (to the nearer side),
    and every shard keeps at least one bin."""
    c = np.asarray(cost, dtype=np.float64)
    nwn = c.size
    if world > nwn or world < 1:
        raise ValueError("bad shard request: nwn=%d world=%d" % (nwn, world))
    if not np.all(c >= 0):
        raise ValueError("negative bin cost")
    if c.sum() <= 0:
        return all_bounds(nwn, world)
    run = np.concatenate([[0.0], np.cumsum(c)])
    cuts = [0]
    for k in range(1, world):
        target = run[-1] * k / world
        j = int(np.searchsorted(run, target))          # run[j-1] < target <= run[j]
        if j > 0 and target - run[j - 1] < run[j] - target:
            j -= 1
        j = min(max(j, cuts[-1] + 1), nwn - (world - k))
        cuts.append(j)
    cuts.append(nwn)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def bin_costs(static, nlayer: int) -> np.ndarray:
    """Work per coarse bin of a problem (plain struct trx_static): every line walks once per step
    whatever its bin (cost ~ lines in the bin's cell), and every bin carries the optical-depth
    and spectrum work of its ray (cost ~ layers).  The weights are the measured per-line and
    per-(bin, layer) times of the demo-sized run."""
    n, nwn = int(static.nlines), int(static.nwn)
    c = np.full(nwn, 0.04 * nlayer)                    # us per bin: optical depth + emission
    if n:
        wl = np.ctypeslib.as_array(static.wl_um, shape=(n,))
        cell = np.floor((1e4 / wl - static.wn_i) / static.wn_d + 0.5).astype(np.int64)
        ok = (cell >= 0) & (cell < nwn)
        c += 3.4e-4 * np.bincount(cell[ok], minlength=nwn)    # us per line and run
    return c


def stitch(parts: Sequence[np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(p) for p in parts])
