"""Multi-GPU jobs: one process per GPU, the wavenumber axis split across ranks
(SURVEY.md section 8e).  The data path has exactly ONE exchange: the gather of the
per-rank spectrum slices at the end -- `Engine.gather` (trx_gather: ncclAllGather on
the handle's stream, communicator from `create_comm`) or, for CPU tests and ranks that
share a device, `gather_spectrum` over torch.distributed.  The per-layer maximum line
strength, the only global quantity of the path (reference extinction.c:399-427), needs
no exchange: every rank computes it from the same small set of candidate lines.

torch.distributed is plumbing here: rendezvous and the 128-byte id broadcast (backend
"nccl" is RCCL on ROCm; "gloo" works for CPU tests).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional

import numpy as np

from .shard import all_bounds, shard_bounds

COMM_ID_BYTES = 128


def create_comm(world: int, rank: int, device: int, broadcast: Optional[Callable[[bytes], bytes]] = None):
    """RCCL communicator for the engine.  `broadcast(payload_from_rank0) -> payload`
    ships the unique id; by default torch.distributed does it."""
    from .engine import hip_library, EngineError
    lib = hip_library()
    lib.trx_comm_unique_id.argtypes = [C.c_void_p]
    lib.trx_comm_unique_id.restype = C.c_int
    lib.trx_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.trx_comm_create.restype = C.c_int
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if rank == 0:
        rc = lib.trx_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(rc, "trx_comm_unique_id")
    payload = bytes(buf.raw)
    if world > 1:
        payload = (broadcast or _torch_broadcast(device))(payload)
    idbuf = C.create_string_buffer(payload, COMM_ID_BYTES)
    comm = C.c_void_p()
    rc = lib.trx_comm_create(idbuf, world, rank, device, C.byref(comm))
    if rc != 0:
        lib.trx_last_error.argtypes, lib.trx_last_error.restype = [C.c_void_p], C.c_char_p
        raise EngineError(rc, "trx_comm_create", (lib.trx_last_error(None) or b"").decode(errors="replace"))
    return comm


def destroy_comm(comm):
    from .engine import hip_library
    lib = hip_library()
    lib.trx_comm_destroy.argtypes = [C.c_void_p]
    lib.trx_comm_destroy.restype = None
    if comm:
        lib.trx_comm_destroy(comm)


def _torch_broadcast(device: int):
    def bc(payload: bytes) -> bytes:
        import torch
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor(list(payload), dtype=torch.uint8, device=("cuda:%d" % device) if on_gpu else "cpu")
        dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())
    return bc


def padded_len(nwn: int, world: int) -> int:
    """Collectives want equal counts: every rank contributes ceil(nwn/world) values."""
    return -(-nwn // world)


def gather_spectrum(local, nwn: int, world: int, rank: int, out=None):
    """All-gather the shard spectra into the full wavenumber grid, in rank order.

    `local` is a torch tensor or numpy array holding this rank's slice in its
    first (hi - lo) entries; it may already be padded to padded_len() (then no
    copy is made).  One collective: all_gather_into_tensor of equal-size pieces."""
    if world == 1:
        lo, hi = shard_bounds(nwn, world, rank)
        return local[: hi - lo]
    import torch
    import torch.distributed as dist
    is_np = not isinstance(local, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local
    m = padded_len(nwn, world)
    if t.numel() != m:
        pad = torch.zeros(m, dtype=t.dtype, device=t.device)
        pad[: t.numel()] = t
        t = pad
    if out is None:
        out = torch.empty(m * world, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    pieces = [out[r * m: r * m + (hi - lo)] for r, (lo, hi) in enumerate(all_bounds(nwn, world))]
    full = torch.cat(pieces)
    return full.numpy() if is_np else full


def sharded_spectrum(run_shard: Callable[[int, int], np.ndarray], nwn: int, world: int, rank: int):
    """Run this rank's shard with `run_shard(lo, hi)` and gather the whole spectrum."""
    lo, hi = shard_bounds(nwn, world, rank)
    local = np.ascontiguousarray(run_shard(lo, hi), dtype=np.float64)
    assert local.shape == (hi - lo,)
    return gather_spectrum(local, nwn, world, rank)
