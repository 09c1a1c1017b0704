"""Run graphs (TRX_RUN_GRAPH=1, opt-in: DESIGN.md section 4): a hinted production run captured
as one HIP graph the second time its plan signature comes by and replayed from then on.  The
spectra must be the queued path's bits -- also when the atmosphere moves under a captured
plan (rays go deeper: the replayed pass is resumed un-captured; shallower: a new signature)."""
import os

import numpy as np
import pytest

from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_replayed_runs_give_the_queued_path_s_bits(tmp_path, solution):
    d = str(tmp_path / "g")
    synth.make_case(d, nlines=30_000, wnlow=2500, wnhigh=2700, wndelt=1.0, wnosamp=2160, nlayers=80,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    dens = np.ctypeslib.as_array(P.atm.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()
    # (the layer maxima of consecutive runs alternate between two buffers, and so do the graphs: a plan
    # is captured when it comes by the second time on the same buffer, replayed from the third)
    scales = [1.0] * 9 + [0.02] * 6 + [30.0] * 6 + [1.0] * 4
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    os.environ["TRX_RUN_GRAPH"] = "1"
    try:
        g = Engine(P.static)
    finally:
        os.environ.pop("TRX_RUN_GRAPH", None)
    q = Engine(P.static)
    try:
        for k, sc in enumerate(scales):
            dens[:] = base * sc
            a, b = g.run(P.atm, P.opts)["spectrum"], q.run(P.atm, P.opts)["spectrum"]
            assert np.array_equal(a, b), (k, sc)
    finally:
        dens[:] = base
        engine.set_log(None)
        g.close(); q.close()
    captured = sum("captured as a graph" in m for m in msgs)
    replayed = sum("graph " in m and "queueing by phase" in m for m in msgs)
    assert captured >= 4 and replayed >= 8, (captured, replayed)
