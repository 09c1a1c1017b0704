"""trx_run_batch: K atmospheres through one batch of handles (a retrieval driver's chains; the reference
calls run_transit once per atmosphere, transit.c:118-122).  Every spectrum of the batch must be the one
trx_run gives for its atmosphere on a single handle, bit for bit -- whichever handle of the batch served
it, whatever it computed before."""
import ctypes as C
import os

import numpy as np
import pytest

from transit_amd import _abi, synth
from transit_amd.engine import Batch, Engine, EngineError
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


def atmospheres(P, k):
    """k variants of the problem's atmosphere: temperatures scaled by up to +-6 %, densities the other way
    (the arrays are kept alive by the returned list)."""
    a = P.atm
    L = P.layer_arrays()
    n = int(a.nlayer)
    keep, atms = [], []
    for j in range(k):
        f = 1.0 + 0.06 * np.sin(1.7 * j + 0.3)
        temp = np.ascontiguousarray(L["temp"] * f)
        dens = np.ascontiguousarray(L["density"] / f)
        keep += [temp, dens]
        b = _abi.TrxAtm()
        C.memmove(C.byref(b), C.byref(a), C.sizeof(_abi.TrxAtm))
        b.temp = temp.ctypes.data_as(_abi.c_double_p)
        b.density = dens.ctypes.data_as(_abi.c_double_p)
        atms.append(b)
    return atms, keep


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_batch_spectra_are_the_single_handle_spectra(tmp_path, solution):
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2900, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=33, ncia=2 if solution == "transit" else 1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    K = 11
    atms, keep = atmospheres(P, K)
    one = Engine(P.static)
    ref = np.stack([one.run(atms[j], P.opts)["spectrum"].copy() for j in range(K)])
    one.close()
    assert len({ref[j].tobytes() for j in range(K)}) == K, "the atmospheres differ, so must their spectra"
    B = Batch(P.static, ways=3)
    for rep in range(3):                       # (the handles' depth hints now come from other atmospheres)
        got = B.run(atms, P.opts)
        assert np.array_equal(got, ref), (rep, float(np.max(np.abs(got - ref))))
    got = B.run(atms[:2], P.opts)               # fewer atmospheres than handles
    assert np.array_equal(got, ref[:2])
    assert B.run([], P.opts).shape == (0, P.nwn)
    B.close()


def test_batch_reports_the_failing_atmosphere(tmp_path):
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=20_000, wnlow=2500, wnhigh=2600, wndelt=1.0, wnosamp=2160, nlayers=60,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    atms, keep = atmospheres(P, 4)
    atms[2].nlayer = 0                          # trx_run refuses this one
    B = Batch(P.static, ways=2)
    with pytest.raises(EngineError) as ei:
        B.run(atms, P.opts)
    assert "atmosphere 2" in str(ei.value)
    atms[2].nlayer = P.atm.nlayer               # and the batch is usable afterwards
    got = B.run(atms, P.opts)
    assert np.all(np.isfinite(got)) and got.shape == (4, P.nwn)
    B.close()
    with pytest.raises(EngineError):
        Batch(P.static, ways=0)
