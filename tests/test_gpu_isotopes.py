"""Isotope count: the reference allocates per isotope without a cap (readlineinfo.c:134-224,
250-278); this path keeps a block's per-isotope tables in LDS and documents its bound
(kMaxIso = 256, trx_device.h).  130 isotopes in 8 databases are golden case `many_isotopes`
(tests/test_gpu_parity.py runs it with the others); here: the same list through a wavenumber
shard (the windowed sweep builds its per-isotope line runs on the device) and the refusal
above the bound."""
import os

import numpy as np
import pytest

from cases import GOLDEN, rel_err
from transit_amd import synth
from transit_amd.engine import Engine, EngineError
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


def test_130_isotopes_sharded_equals_unsharded():
    P = Problem.from_cfg(os.path.join(GOLDEN, "many_isotopes", "case.cfg"))
    assert P.static.niso == 130
    e = Engine(P.static)
    full = e.run(P.atm, P.opts, debug=("e", "computed"))
    e.close()
    for lo, hi in ((0, 9), (9, P.nwn)):
        P.set_shard(lo, hi)
        try:
            e = Engine(P.static)
            part = e.run(P.atm, P.opts, debug=("e", "computed"))
            e.close()
        finally:
            P.set_shard(0, P.nwn)
        assert np.array_equal(part["spectrum"], full["spectrum"][lo:hi])
        sw = full["computed"].astype(bool) & part["computed"].astype(bool)
        assert np.array_equal(part["e"][sw], full["e"][sw][:, lo:hi])


def test_130_isotopes_wide_profiles_against_oracle(tmp_path):
    """The same databases on a fine grid: the two-kernel form (k_group_sweep's line runs, k_accumulate*)."""
    import oracle_lib as ol
    d = str(tmp_path / "fine")
    synth.make_case(d, dbs=synth.many_isotope_dbs(lines_per_iso=12, wn_lo=2500.0, wn_hi=2501.0), wnlow=2500, wnhigh=2501,
                    wndelt=0.002, wnosamp=1, nlayers=10)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    hip = Engine(P.static)
    got = hip.run(P.atm, P.opts, debug=True)
    hip.close()
    ora = ol.OracleEngine(P.static)
    ref = ora.run(P.atm, P.opts, debug=True)
    ora.close()
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    sw = got["computed"].astype(bool) & ref["computed"].astype(bool)
    assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-9


def test_more_isotopes_than_the_bound_are_refused(tmp_path):
    d = str(tmp_path / "many")
    synth.make_case(d, dbs=synth.many_isotope_dbs(niso_total=264, lines_per_iso=3), wnlow=2500, wnhigh=2520, nlayers=8)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    with pytest.raises(EngineError) as ei:
        Engine(P.static)
    assert ei.value.code == -6          # TRX_E_UNSUPPORTED
