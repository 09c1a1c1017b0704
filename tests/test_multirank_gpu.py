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
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0      # BASELINE.json's metric: the one demo split N ways
    assert out["config"]["n_wn"] == 201 and out["config"]["n_lines"] == 30000
    assert {"roofline", "metric", "unit", "ms_per_step"} <= set(out)
    # the rehearsal's own check (bench.py exits non-zero when it fails): the two ranks' stitched spectrum is the
    # one-GPU spectrum bit for bit; and every rank's share of the time is in the line
    assert out["config"]["rehearse_stitched_equals_one_gpu_bitwise"] is True
    assert [r["rank"] for r in out["config"]["per_rank"]] == [0, 1] and sum(r["bins"] for r in out["config"]["per_rank"]) == 201


@pytest.mark.gpu
def test_bench_secondary_measurements_are_guarded():
    """The N>1 extras (other scaling mode, configs[4] split N ways) ride along under a time budget that
    rank 0 decides for all: with no budget they are skipped on every rank and the headline is printed."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", TRANSIT_BENCH_EXTRAS_BUDGET_S="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "2", "--rehearse", "--lines", "30000", "--layers", "40", "--wnhigh", "2700"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["value"] > 0 and "skipped" in out["config"]["weak"] and "skipped" in out["config"]["c5_strong"]


@pytest.mark.gpu
def test_rccl_two_ranks_on_two_devices():
    """The one exchange of the path over RCCL with more than one rank -- trx_gather from two
    processes, and `transit_hip --gpus 2` (two threads of one process, each with its own
    ncclCommInitRank) -- against the unsharded result, bit for bit.  Needs two devices: skipped on
    the one-GPU test boxes (where ranks share the device and the slices are joined in host memory)."""
    from transit_amd.engine import device_count
    if device_count() < 2:
        pytest.skip("one device: RCCL refuses two ranks on it")
    import shutil
    import tempfile
    import numpy as np
    from cases import GOLDEN
    from transit_amd import build
    root = os.path.dirname(HERE)
    # (a) bench.py's path: torch.distributed ranks, trx_comm_create + Engine.gather
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-extras", "--lines", "30000", "--layers", "40", "--wnhigh", "2700"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    # (b) the command: two handles on two host threads, one ncclAllGather
    exe = build.build_cli() or build.lib_path("transit_hip")
    outs = {}
    for n in (1, 2):
        work = os.path.join(tempfile.mkdtemp(), "c")
        shutil.copytree(os.path.join(GOLDEN, "eclipse_small"), work)
        os.remove(os.path.join(work, "spectrum.dat"))
        q = subprocess.run([exe, "-c", "case.cfg", "--gpus", str(n)], cwd=work, capture_output=True, text=True, timeout=300)
        assert q.returncode == 0, q.stderr
        assert "joined in host memory" not in q.stderr
        outs[n] = open(os.path.join(work, "spectrum.dat")).read()
    assert outs[1] == outs[2]
