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
import numpy as np, torch.distributed as dist
import oracle_lib as ol
from cases import golden
from transit_amd.dist import sharded_spectrum
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = golden(%(case)r)
eng = ol.OracleEngine(g.problem.static)
full = eng.run(g.problem.atm, g.problem.opts)["spectrum"]
got = sharded_spectrum(lambda lo, hi: full[lo:hi], g.problem.nwn, world, rank)
assert np.array_equal(got, full), "rank %%d: gathered spectrum differs" %% rank
if rank == 0:
    np.save(%(out)r, got)
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


@pytest.mark.parametrize("case", ["eclipse_small", "transit_small"])
def test_two_rank_gloo_gather(tmp_path, case):
    out = str(tmp_path / "spec.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "case": case, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    from cases import golden
    import oracle_lib as ol
    g = golden(case)
    full = ol.OracleEngine(g.problem.static).run(g.problem.atm, g.problem.opts)["spectrum"]
    assert np.array_equal(np.load(out), full)
