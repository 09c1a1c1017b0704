"""k_accumulate_rows (wide profiles on a grid without oversampling: rows of the Voigt table
staged in LDS, trx_rows.hip.h) against k_accumulate_wide (the general form): the same sums
in the same order, so the same BITS -- extinction, optical depth, spectrum, counters --
on the whole grid, on shards that start inside a tile, with a threshold that drops groups,
and on a sparse list whose runs are single groups."""
import os

import numpy as np
import pytest

from transit_amd import synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu

C5 = dict(wndelt=9.667e-4, wnosamp=1, nlayers=150, solution="eclipse", toomuch=10.0)


def run_form(P, var, profile, debug):
    if var:
        os.environ[var[0]] = var[1]
    try:
        e = Engine(P.static)
    finally:
        if var:
            os.environ.pop(var[0], None)
    P.opts.profile = profile
    try:
        got = e.run(P.atm, P.opts, debug=debug)
        st = e.stats()
    finally:
        P.opts.profile = 0
        e.close()
    return got, st


def both_forms(P, debug=("e", "tau", "last", "computed")):
    """rows staged with tiles of 256 bins everywhere, of 512 bins everywhere, the general form; each
    with the counters of its counting instantiation (a run of its own: trx_opts.profile = 2), which
    must give the same bits."""
    out = []
    for var in (("TRX_ROWS_M8_FROM", "1000000000"), ("TRX_ROWS_M8_FROM", "1"), ("TRX_NO_ROW_STAGING", "1")):
        got, _ = run_form(P, var, 0, debug)
        cnt, st = run_form(P, var, 2, debug)
        assert_same_bits((got, {}), (cnt, {}))
        out.append((got, st))
    return out


def assert_same_bits(a, b):
    (ga, sa), (gb, sb) = a, b
    sw = slice(None)
    if "computed" in ga:
        assert np.array_equal(ga["computed"], gb["computed"])
        sw = ga["computed"].astype(bool)
    for k in ga:
        if k == "e":
            assert np.array_equal(ga[k][sw], gb[k][sw]), k
        else:
            assert np.array_equal(ga[k], gb[k]), k
    for k in ("sum_bins", "neval", "nskip", "nadd", "layers_swept"):
        if k in sa:
            assert sa[k] == sb[k], k


@pytest.mark.parametrize("nlines,ethresh", [(24_000, 1e-50), (24_000, 1e-4), (300, 1e-50)])
def test_rows_form_gives_the_bits_of_the_general_form(tmp_path, nlines, ethresh):
    d = str(tmp_path / "win")
    synth.make_case(d, nlines=nlines, wnlow=3000, wnhigh=3024, seed=5, ethresh=ethresh, **C5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn > 24_000
    rows, rows64, wide = both_forms(P)
    assert rows[1]["sum_bins"] > 0
    assert_same_bits(rows, wide)
    assert_same_bits(rows64, wide)
    # shards that begin and end inside a tile: the same bits as the unsharded run
    for lo, hi in ((1000, 9001), (9001, P.nwn)):
        P.set_shard(lo, hi)
        try:
            part = both_forms(P, debug=("e", "computed"))
        finally:
            P.set_shard(0, P.nwn)
        assert_same_bits(part[0], part[2])
        assert_same_bits(part[1], part[2])
        sw = rows[0]["computed"].astype(bool)
        assert np.array_equal(part[0][0]["e"][sw], rows[0]["e"][sw][:, lo:hi])
        assert np.array_equal(part[0][0]["spectrum"], rows[0]["spectrum"][lo:hi])


def test_rows_form_on_the_high_resolution_golden():
    here = os.path.dirname(os.path.abspath(__file__))
    P = Problem.from_cfg(os.path.join(here, "golden", "highres_fine", "case.cfg"))
    rows, rows64, wide = both_forms(P)
    assert_same_bits(rows, wide)
    assert_same_bits(rows64, wide)
