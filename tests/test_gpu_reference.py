"""The production path against the COMPILED REFERENCE at benchmark-like sizes, as a pass/fail of the GPU suite
(bench.py only prints the figure: cpu_baseline.gpu_vs_reference_max_rel): CH4-demo-shaped inputs of 10^5 and 10^6
lines, 100 layers, both geometries; `oracle/_ref/transit` (built from /root/reference by oracle/Makefile, shipped
with the tree) writes spectrum.dat and toomuch.dat, the handle's HINTED run -- the one every timed step of the
bench is: walks + k_ray_tail -- must give the spectrum to the reference's print precision (2e-8) and stop every
ray at the same layer.  Skipped where the reference binary is absent."""
import os
import subprocess

import numpy as np
import pytest

from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "transit")


@pytest.mark.parametrize("nlines,solution", [(100_000, "eclipse"), (100_000, "transit"), (1_000_000, "eclipse"), (1_000_000, "transit")])
def test_hinted_production_run_against_the_reference_binary(tmp_path, nlines, solution):
    if not (os.path.exists(REF) and os.access(REF, os.X_OK)):
        pytest.skip("oracle/_ref/transit is not built (needs /root/reference at build time)")
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=nlines, wnlow=2500.0, wnhigh=5000.0, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution=solution, toomuch=10.0, ethresh=1e-50, nwidth=20.0, raygrid="0 20 40 60 80",
                    ncia=2 if solution == "transit" else 1, seed=1234)
    p = subprocess.run([REF, "-c", "case.cfg"], cwd=d, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    ref_spec = np.loadtxt(os.path.join(d, "spectrum.dat"), comments="#")[:, 1]
    ref_last = np.loadtxt(os.path.join(d, "toomuch.dat"), comments="#", skiprows=2)[:, 3].astype(np.int64)

    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    eng = Engine(P.static)
    eng.run(P.atm, P.opts)                                        # unhinted: the handle learns the depth
    got = eng.run(P.atm, P.opts, debug=("last",))                 # (debug copies only: the kernels are the production ones)
    prod = eng.run(P.atm, P.opts)                                 # hinted production run: walks + k_ray_tail
    engine.set_log(None)
    eng.close()
    assert any("ray tail over" in m for m in msgs), "the hinted run did not take the production path (k_ray_tail)"
    if solution == "eclipse":       # the deep step (17 layers, 8-bin frames on a dense list) is k_line_walk_lanes: checked HERE against the reference
        assert any("walk: lanes = lines" in m for m in msgs), "the deep step did not run k_line_walk_lanes"
    assert np.array_equal(prod["spectrum"], got["spectrum"])
    assert len(ref_spec) == P.nwn
    assert np.array_equal(got["last"], ref_last), "rays stop at other layers than the reference's"
    rel = np.max(np.abs(prod["spectrum"] / ref_spec - 1.0))
    assert rel < 2e-8, "spectrum differs from the reference's by %g" % rel


def test_three_species_200_layers_against_the_reference_binary(tmp_path):
    """BASELINE configs[2]'s shape -- three line databases (H2O, CH4, CO: six isotopes), 333.33-10000 cm-1 at 1 cm-1,
    200 layers -- with 10^5 lines per database: the compiled reference's spectrum and stopping layers against the
    hinted production run (a plan of three walk steps: the step kernels behind them, not k_ray_tail)."""
    if not (os.path.exists(REF) and os.access(REF, os.X_OK)):
        pytest.skip("oracle/_ref/transit is not built (needs /root/reference at build time)")
    d = str(tmp_path / "c3")
    synth.make_case(d, dbs=synth.three_species_dbs(100_000, 333.33, 10000), wnlow=333.33, wnhigh=10000, wndelt=1.0, wnosamp=2160,
                    nlayers=200, solution="eclipse", toomuch=10.0, ethresh=1e-50, nwidth=20.0, raygrid="0 20 40 60 80", ncia=1)
    p = subprocess.run([REF, "-c", "case.cfg"], cwd=d, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    ref_spec = np.loadtxt(os.path.join(d, "spectrum.dat"), comments="#")[:, 1]
    ref_last = np.loadtxt(os.path.join(d, "toomuch.dat"), comments="#", skiprows=2)[:, 3].astype(np.int64)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn == 9667 and P.nlayer == 200 and P.static.niso == 6
    eng = Engine(P.static)
    eng.run(P.atm, P.opts)                                        # unhinted: the handle learns the depth
    got = eng.run(P.atm, P.opts, debug=("last",))
    prod = eng.run(P.atm, P.opts)                                 # hinted production run
    st = eng.stats()
    eng.close()
    assert st["walk_steps"] >= 2
    assert np.array_equal(prod["spectrum"], got["spectrum"])
    assert np.array_equal(got["last"], ref_last), "rays stop at other layers than the reference's"
    rel = np.max(np.abs(prod["spectrum"] / ref_spec - 1.0))
    assert rel < 2e-8, "spectrum differs from the reference's by %g" % rel
