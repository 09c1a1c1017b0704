"""One rank of a sharded job on a SHARED GPU (launched by tests/test_multirank_gpu.py through
torch.distributed.run, backend gloo).  RCCL refuses two ranks on one device, so the one
exchange of the path -- the gather of the slices -- goes over gloo here; everything else is
the engine as every rank of an N-GPU job runs it: its own line window, its own depth hint and
stop/resume decisions, layer maxima from the shared candidate set.  Rank 0 checks the
stitched spectra against an unsharded handle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from cases import golden, rel_err                     # noqa: E402
from transit_amd import dist as tdist                 # noqa: E402
from transit_amd.engine import Engine                 # noqa: E402
from transit_amd.shard import shard_bounds            # noqa: E402


def main():
    case = sys.argv[1]
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    P = golden(case).problem
    nwn = P.nwn
    a = P.atm
    dens = np.ctypeslib.as_array(a.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()
    # first run, hinted, more transparent (deeper: resume below the hint), back, more opaque, ...
    scales = tuple(float(x) for x in sys.argv[2].split(","))

    refs = []
    if rank == 0:                                      # the unsharded answer, fresh handle per atmosphere
        for sc in scales:
            dens[:] = base * sc
            e = Engine(P.static)
            refs.append(e.run(P.atm, P.opts, debug=("last",)))
            e.close()
        dens[:] = base

    lo, hi = shard_bounds(nwn, world, rank)
    P.set_shard(lo, hi)
    eng = Engine(P.static)
    mpad = tdist.padded_len(nwn, world)
    ok = True
    for k, sc in enumerate(scales):
        dens[:] = base * sc
        out = eng.run(P.atm, P.opts, debug=("last",))
        mine = torch.zeros(mpad, dtype=torch.float64)
        mine[: hi - lo] = torch.from_numpy(out["spectrum"])
        lastp = torch.full((mpad,), -7, dtype=torch.int64)
        lastp[: hi - lo] = torch.from_numpy(out["last"])
        full = tdist.gather_spectrum(mine, nwn, world, rank).numpy()
        lasts = [torch.empty_like(lastp) for _ in range(world)]
        dist.all_gather(lasts, lastp)
        if rank == 0:
            last_full = np.concatenate([lasts[r][: shard_bounds(nwn, world, r)[1] - shard_bounds(nwn, world, r)[0]].numpy()
                                        for r in range(world)])
            err = rel_err(full, refs[k]["spectrum"])
            same_last = np.array_equal(last_full, refs[k]["last"])
            print("run %d scale %g: spectrum rel err %.2e, last equal %s, depth %d" %
                  (k, sc, err, same_last, int(refs[k]["last"].max()) + 1), flush=True)
            ok &= err < 1e-11 and same_last
    dens[:] = base
    eng.close()
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    if rank == 0:
        print("MULTIRANK OK" if ok else "MULTIRANK MISMATCH", flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
