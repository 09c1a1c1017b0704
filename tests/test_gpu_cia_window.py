"""The CIA wavenumber spline solved for the table rows a run's wavenumbers bracket (128 rows to
spare on either side) instead of the whole table (k_cia_layers): what the tridiagonal sweeps carry
from row to row shrinks by the pivots' ratio every row, so inside the margin the second derivatives
are the whole table's doubles -- e_cs and the spectrum of a shard of a wide band must be the same
BITS as with TRX_CIA_WINDOW=0, in both geometries, on shards at the start, in the middle and at the
end of the table, with one and with two tables."""
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
            try:
                b = Engine(P.static)
            finally:
                os.environ.pop("TRX_CIA_WINDOW", None)
            try:
                for rep in range(2):
                    ra = a.run(P.atm, P.opts, debug=("e_cs", "last", "tau"))
                    rb = b.run(P.atm, P.opts, debug=("e_cs", "last", "tau"))
                    assert np.array_equal(ra["e_cs"], rb["e_cs"]), (k, rep)
                    assert np.array_equal(ra["last"], rb["last"]), (k, rep)
                    assert np.array_equal(ra["spectrum"], rb["spectrum"]), (k, rep)
                    assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
                windows += 1
            finally:
                a.close(); b.close()
        finally:
            P.set_shard(0, P.nwn)
    assert windows == 4
