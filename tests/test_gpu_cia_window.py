"""The CIA wavenumber spline solved for the table rows a run's wavenumbers bracket (128 rows to
spare on either side) instead of the whole table (k_cia_layers): what the tridiagonal sweeps carry
from row to row shrinks by the pivots' ratio every row, so inside the margin the second derivatives
are the whole table's doubles -- e_cs and the spectrum of a shard of a wide band must be the same
BITS as with TRX_CIA_WINDOW=0, in both geometries, on shards at the start, in the middle and at the
end of the table, with one and with two tables.  The same argument cuts the rows of a window into
segments of 128 that are solved side by side, each with margins of its own (the sweeps are a chain
of dependent steps on two waves: the longest kernel of the CIA queue): the default engine here has
windows AND segments, the one it is compared with (TRX_CIA_WINDOW=0, TRX_CIA_SEGMENTS=0) solves
every table whole in one sweep, as the reference does."""
import os

import numpy as np
import pytest

from transit_amd import synth
from transit_amd.engine import Engine
from transit_amd.host import Problem
from transit_amd.shard import all_bounds

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solution,ncia", [("eclipse", 1), ("transit", 2)])
def test_windowed_cia_spline_is_the_whole_table_s(tmp_path, solution, ncia):
    d = str(tmp_path / "w")
    synth.make_case(d, nlines=60_000, wnlow=2000, wnhigh=12000, wndelt=1.0, wnosamp=2160, nlayers=50,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=5, ncia=ncia)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    windows = 0
    for k, (lo, hi) in enumerate(all_bounds(P.nwn, 7)):
        if k in (2, 4, 5):
            continue
        P.set_shard(lo, hi)
        try:
            a = Engine(P.static)
            os.environ["TRX_CIA_WINDOW"] = "0"
            os.environ["TRX_CIA_SEGMENTS"] = "0"
            try:
                b = Engine(P.static)
                os.environ.pop("TRX_CIA_WINDOW", None)
                c = Engine(P.static)                  # windows, one segment each
            finally:
                os.environ.pop("TRX_CIA_WINDOW", None)
                os.environ.pop("TRX_CIA_SEGMENTS", None)
            try:
                for rep in range(2):
                    ra = a.run(P.atm, P.opts, debug=("e_cs", "last", "tau"))
                    rb = b.run(P.atm, P.opts, debug=("e_cs", "last", "tau"))
                    rc = c.run(P.atm, P.opts, debug=("e_cs", "last", "tau"))
                    assert np.array_equal(ra["e_cs"], rb["e_cs"]), (k, rep)
                    assert np.array_equal(rc["e_cs"], rb["e_cs"]), (k, rep)
                    assert np.array_equal(ra["last"], rb["last"]), (k, rep)
                    assert np.array_equal(ra["spectrum"], rb["spectrum"]), (k, rep)
                    assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
                windows += 1
            finally:
                a.close(); b.close(); c.close()
        finally:
            P.set_shard(0, P.nwn)
    assert windows == 4


@pytest.mark.parametrize("solution,ncia", [("eclipse", 1), ("transit", 2)])
def test_segmented_cia_spline_on_the_whole_grid(tmp_path, solution, ncia):
    """No shard, no window to speak of (the run's wavenumbers cover most of the table): the segments alone
    against one sweep per table."""
    d = str(tmp_path / "s")
    synth.make_case(d, nlines=30_000, wnlow=2000, wnhigh=12000, wndelt=1.0, wnosamp=2160, nlayers=40,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=8, ncia=ncia)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a = Engine(P.static)
    os.environ["TRX_CIA_SEGMENTS"] = "0"
    try:
        b = Engine(P.static)
    finally:
        os.environ.pop("TRX_CIA_SEGMENTS", None)
    try:
        ra = a.run(P.atm, P.opts, debug=("e_cs", "last"))
        rb = b.run(P.atm, P.opts, debug=("e_cs", "last"))
        assert np.array_equal(ra["e_cs"], rb["e_cs"])
        assert np.array_equal(ra["spectrum"], rb["spectrum"])
        assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
    finally:
        a.close(); b.close()
