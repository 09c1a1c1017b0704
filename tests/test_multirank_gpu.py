"""A real sharded job: N processes, N shards, one (shared) GPU, exchanges over the host
transport.  See tests/multirank_gpu_worker.py."""
import os
import subprocess
import sys

import pytest

from cases import free_port

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,scales", [("coadd_thresh", 2, "1,1,0.03,1,20,0.03"),
                                               ("transit_small", 3, "1,1,0.3,1,3,0.3")])
def test_sharded_job_matches_the_unsharded_handle(case, world, scales):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "multirank_gpu_worker.py"), case, scales]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "MULTIRANK OK" in p.stdout, p.stdout[-3000:]


@pytest.mark.gpu
def test_bench_multirank_code_path_rehearsal():
    """bench.py's N>1 branch (rank-0 workload + barrier, shards, padded gather, max-over-ranks
    timing, one JSON line from rank 0) with two ranks sharing the GPU (--rehearse)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "2", "--rehearse", "--no-extras", "--lines", "30000",
           "--layers", "40", "--wnhigh", "2700"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0      # the one run, split two ways
    assert out["config"]["n_wn"] == 201 and out["config"]["n_lines"] == 30000
    assert {"roofline", "metric", "unit", "ms_per_step"} <= set(out)
