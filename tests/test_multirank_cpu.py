"""N > 1 path on CPU: two gloo ranks, the wavenumber axis split between them,
each rank produces its shard (here with the CPU oracle standing in for the
engine) and the gathered spectrum must equal the single-process one."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cases import free_port

from transit_amd.shard import all_bounds, shard_bounds, stitch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as ol
from cases import golden
from transit_amd.shard import balanced_bounds, bin_costs
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
P = golden(%(case)r).problem
nwn = P.nwn
# what every rank of a sharded job does (bench.py sharded_run, transit_hip --gpus N): bins cut by
# cost, a handle for the rank's OWN shard (trx_static.wn_lo / wn_hi), its spectrum slice padded to
# the longest shard, one all-gather of equal counts, the slices cut back out in rank order
bounds = balanced_bounds(bin_costs(P.static, P.nlayer), world)
lo, hi = bounds[rank]
P.set_shard(lo, hi)
eng = ol.OracleEngine(P.static)
part = eng.run(P.atm, P.opts, debug=("last", "tau"))
eng.close()
assert part["spectrum"].shape == (hi - lo,) and part["tau"].shape == (hi - lo, P.nlayer)
mpad = max(h - l for l, h in bounds)
mine = torch.zeros(mpad, dtype=torch.float64); mine[: hi - lo] = torch.from_numpy(part["spectrum"])
gathered = torch.zeros(mpad * world, dtype=torch.float64)
dist.all_gather_into_tensor(gathered, mine)
g = gathered.numpy()
full = np.concatenate([g[k * mpad: k * mpad + (h - l)] for k, (l, h) in enumerate(bounds)])
np.save(%(out)r %% rank, full)
np.save(%(out)r %% (10 + rank), part["last"])
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_bounds_partition():
    for nwn in (2, 61, 2501, 10_000_019):
        for world in (1, 2, 3, 8):
            if world > nwn:
                continue
            b = all_bounds(nwn, world)
            assert b[0][0] == 0 and b[-1][1] == nwn
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 8, 0)
    assert np.array_equal(stitch([np.arange(3), np.arange(3, 5)]), np.arange(5))


@pytest.mark.parametrize("case,world", [("eclipse_small", 2), ("transit_small", 2), ("coadd_thresh", 3)])
def test_gloo_ranks_run_their_own_shards(tmp_path, case, world):
    """Every rank runs its OWN cost-balanced shard (the CPU oracle standing in for the engine behind
    the same trx_static.wn_lo / wn_hi) and the gathered slices are the unsharded spectrum, bit for
    bit, on every rank."""
    out = str(tmp_path / "spec%d.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "case": case, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    from cases import golden
    import oracle_lib as ol
    from transit_amd.shard import balanced_bounds, bin_costs
    g = golden(case)
    ref = ol.OracleEngine(g.problem.static).run(g.problem.atm, g.problem.opts, debug=("last",))
    bounds = balanced_bounds(bin_costs(g.problem.static, g.problem.nlayer), world)
    assert len(set(h - l for l, h in bounds)) > 1 or world == 1 or g.problem.nwn % world == 0     # (unequal slices: the padding matters)
    for r in range(world):
        got = np.load(out % r)
        # The reference sweeps a layer when the first ray needs it (tau.c:246-270) and the bottom
        # parabola of a ray reads one layer further down: the FIRST ray of a shard may meet a zero
        # there where the unsharded run had that layer from an earlier wavenumber (SURVEY appendix A,
        # item 9; 5e-15 on transit_small).  Every other bin: the same bits.
        differs = np.nonzero(got != ref["spectrum"])[0]
        assert set(differs) <= {lo for lo, _ in bounds[1:]}, (r, differs)
        assert np.max(np.abs(got / ref["spectrum"] - 1)) < 1e-12, r
        lo, hi = bounds[r]
        assert np.array_equal(np.load(out % (10 + r)), ref["last"][lo:hi]), r
