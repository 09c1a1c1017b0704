"""k_line_walk_packed (steps of at most 32 layers: several line ranges per wavefront, trx_walk.hip.h)
against k_line_walk (one range per wave): the same lines in the same order into the same partial
records, so the same BITS -- extinction, optical depth, spectrum -- on the goldens whose walks
have few layers, on a demo-shaped case whose second step has 17, on a shard, and on an atmosphere
of 5 layers (12 ranges per wave)."""
import os

import numpy as np
import pytest

from cases import GOLDEN
from transit_amd import synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


def both(P, runs=2, debug=("e", "tau", "last", "computed")):
    out = []
    for var in ("TRX_PACKED_MAX_LAYERS", "TRX_NO_PACKED_WALK"):       # packed for every step of up to 32 layers / never
        os.environ[var] = "32"
        try:
            e = Engine(P.static)
        finally:
            os.environ.pop(var, None)
        res = [e.run(P.atm, P.opts, debug=debug) for _ in range(runs)]      # unhinted, then hinted (another step plan)
        res.append(e.run(P.atm, P.opts))                                      # and a production run (tile skipping on)
        e.close()
        out.append(res)
    return out


def assert_same(a, b):
    for ra, rb in zip(a, b):
        sw = ra["computed"].astype(bool) if "computed" in ra else slice(None)
        for k in ra:
            if k == "e":
                assert np.array_equal(ra[k][sw], rb[k][sw]), k
            else:
                assert np.array_equal(ra[k], rb[k]), k


@pytest.mark.parametrize("case", ["eclipse_small", "transit_small", "coadd_thresh", "cloud_scatter", "multi_species",
                                  "many_isotopes", "resample_transit", "midres_os4"])
def test_packed_walk_on_goldens(case):
    P = Problem.from_cfg(os.path.join(GOLDEN, case, "case.cfg"))
    a, b = both(P)
    assert_same(a, b)


@pytest.mark.parametrize("nlayers,solution", [(100, "eclipse"), (100, "transit"), (5, "eclipse"), (23, "transit")])
def test_packed_walk_demo_shape_and_few_layers(tmp_path, nlayers, solution):
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2800, wndelt=1.0, wnosamp=2160, nlayers=nlayers,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=11, ncia=2 if solution == "transit" else 1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a, b = both(P)
    assert_same(a, b)
    P.set_shard(37, 211)                       # a shard: only the ranges that reach it are launched
    try:
        sa, sb = both(P, runs=1)
    finally:
        P.set_shard(0, P.nwn)
    assert_same(sa, sb)
    assert np.array_equal(sa[-1]["spectrum"], a[-1]["spectrum"][37:211])
