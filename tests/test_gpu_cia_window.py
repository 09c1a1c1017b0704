"""The CIA wavenumber spline (second derivatives of every layer's column, crosssec.c:354-428) on the device.

Sweeps (k_cia_layers, the reference's two recurrences): solved for the table rows a run's wavenumbers bracket
(128 rows to spare on either side) instead of the whole table -- what the sweeps carry from row to row shrinks
by the pivots' ratio every row, so inside the margin the second derivatives are the whole table's doubles --
and, by the same argument, in segments of 128 rows side by side.  e_cs and the spectrum of a shard of a wide
band must be the same BITS as with TRX_CIA_WINDOW=0 TRX_CIA_SEGMENTS=0.

Sums (k_cia_v, k_cia_z, the default): every row a sum of 48 terms with table-constant weights -- no chain.  A
row depends on its neighbours only, so windows change nothing by construction (bit for bit again); against the
sweeps the rounding differs in the last place: 1e-13 of the largest value, the tolerance e_cs has against the
oracle."""
import os

import numpy as np
import pytest

from transit_amd import synth
from transit_amd.engine import Engine
from transit_amd.host import Problem
from transit_amd.shard import all_bounds

pytestmark = pytest.mark.gpu


def _engine(static, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(static)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


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
        engines = []
        try:
            a = _engine(P.static); engines.append(a)                                           # sums, windows
            a2 = _engine(P.static, TRX_CIA_WINDOW=0); engines.append(a2)                        # sums, whole tables
            b = _engine(P.static, TRX_CIA_SUMS=0, TRX_CIA_WINDOW=0, TRX_CIA_SEGMENTS=0); engines.append(b)      # the reference's sweeps over whole tables
            c = _engine(P.static, TRX_CIA_SUMS=0); engines.append(c)                            # sweeps: windows, segments
            c1 = _engine(P.static, TRX_CIA_SUMS=0, TRX_CIA_SEGMENTS=0); engines.append(c1)      # sweeps: windows, one segment each
            for rep in range(2):
                ra, ra2, rb, rc, rc1 = (e.run(P.atm, P.opts, debug=("e_cs", "last", "tau")) for e in (a, a2, b, c, c1))
                for r in (rc, rc1):                                  # the sweeps, however cut: the same bits
                    assert np.array_equal(r["e_cs"], rb["e_cs"]), (k, rep)
                    assert np.array_equal(r["spectrum"], rb["spectrum"]), (k, rep)
                assert np.array_equal(ra["e_cs"], ra2["e_cs"]), (k, rep)        # the sums: windows change nothing
                assert np.array_equal(ra["spectrum"], ra2["spectrum"]), (k, rep)
                scale = np.abs(rb["e_cs"]).max()
                assert np.abs(ra["e_cs"] - rb["e_cs"]).max() <= 1e-13 * scale, (k, rep)     # sums against sweeps
                assert np.array_equal(ra["last"], rb["last"]), (k, rep)
                np.testing.assert_allclose(ra["spectrum"], rb["spectrum"], rtol=1e-11, atol=0)
                assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
            windows += 1
        finally:
            for e in engines:
                e.close()
            P.set_shard(0, P.nwn)
    assert windows == 4


@pytest.mark.parametrize("solution,ncia", [("eclipse", 1), ("transit", 2)])
def test_segmented_cia_spline_on_the_whole_grid(tmp_path, solution, ncia):
    """No shard, no window to speak of (the run's wavenumbers cover most of the table): the segments alone
    against one sweep per table, and the sums against both."""
    d = str(tmp_path / "s")
    synth.make_case(d, nlines=30_000, wnlow=2000, wnhigh=12000, wndelt=1.0, wnosamp=2160, nlayers=40,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=8, ncia=ncia)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a = _engine(P.static, TRX_CIA_SUMS=0)
    b = _engine(P.static, TRX_CIA_SUMS=0, TRX_CIA_SEGMENTS=0)
    c = _engine(P.static)
    try:
        ra = a.run(P.atm, P.opts, debug=("e_cs", "last"))
        rb = b.run(P.atm, P.opts, debug=("e_cs", "last"))
        rc = c.run(P.atm, P.opts, debug=("e_cs", "last"))
        assert np.array_equal(ra["e_cs"], rb["e_cs"])
        assert np.array_equal(ra["spectrum"], rb["spectrum"])
        assert np.abs(rc["e_cs"] - rb["e_cs"]).max() <= 1e-13 * np.abs(rb["e_cs"]).max()
        assert np.array_equal(rc["last"], rb["last"])
        assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
    finally:
        a.close(); b.close(); c.close()


def test_irregular_cia_grid(tmp_path):
    """A table whose wavenumber spacing jumps by factors of a hundred (clusters of close rows far apart): the sweeps'
    row-to-row factors stay below 0.3 there too -- trx_create's check of the 48th term passes (said in the debug log;
    a table that failed it would keep the two sweeps) -- and e_cs agrees with the oracle and with the sweeps as on a
    regular grid."""
    import oracle_lib as ol
    from cases import rel_err
    from transit_amd import engine
    d = str(tmp_path / "i")
    synth.make_case(d, nlines=20_000, wnlow=2500, wnhigh=3500, wndelt=1.0, wnosamp=2160, nlayers=30,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=13, ncia=2)
    # the second table on an irregular grid: clusters of close rows far apart
    wn, t, cs = synth.synth_cia(2400.0, 3600.0)
    rng = np.random.default_rng(3)
    parts = [c + np.sort(rng.uniform(0.0, 0.5, 6)) for c in np.arange(2400.0, 3600.0, 60.0)]
    wi = np.unique(np.concatenate(parts))
    ci = np.stack([np.interp(wi, wn, cs[:, k]) for k in range(cs.shape[1])], axis=1)
    synth.write_cia(os.path.join(d, "cia_h2he.dat"), ["H2", "He"], wi, t[4:], 0.3 * ci[:, 4:])
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    try:
        a = _engine(P.static)
    finally:
        engine.set_log(None)
    b = _engine(P.static, TRX_CIA_SUMS=0)
    ora = ol.OracleEngine(P.static)
    try:
        how = [m for m in msgs if "second derivatives by" in m]
        assert len(how) == 2 and all("sums per row" in m for m in how), how
        ra = a.run(P.atm, P.opts, debug=("e_cs", "last"))
        rb = b.run(P.atm, P.opts, debug=("e_cs", "last"))
        ro = ora.run(P.atm, P.opts, debug=("e_cs", "last"))
        assert rel_err(ra["e_cs"], ro["e_cs"]) < 1e-13 and rel_err(rb["e_cs"], ro["e_cs"]) < 1e-13
        assert rel_err(ra["e_cs"], rb["e_cs"]) < 1e-13
        assert np.array_equal(ra["last"], ro["last"])
        assert np.all(np.isfinite(ra["e_cs"])) and ra["e_cs"].max() > 0
    finally:
        a.close(); b.close(); ora.close()
