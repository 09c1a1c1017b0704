"""Seeded random problems: the HIP path against the CPU oracle over corners of the parameter
space no hand-written case sits on (oversampling 1..60, 3..70 layers, 0..5000 lines in bands
of 2..40 cm-1, thresholds from 1e-50 to 1e-3, both geometries, 0..2 CIA tables, repeated
runs on one handle so that the depth-hint plan is exercised too)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from cases import rel_err
from transit_amd import synth
from transit_amd.engine import Engine, EngineError
from transit_amd.host import Problem
from tolerances import DEBUG_KEYS, assert_tau_close

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    solution = "eclipse" if rng.random() < 0.6 else "transit"
    wnlow = float(rng.choice([400.0, 2500.0, 4000.0, 9000.0]))
    width = float(rng.choice([2.0, 7.0, 20.0, 40.0]))
    wndelt = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    osamp = int(rng.choice([1, 2, 7, 60, 2160])) if wndelt >= 0.5 else int(rng.choice([1, 2, 5]))
    nlines = int(rng.choice([17, 300, 2000, 5000]))          # (empty and one-line lists: test_gpu_properties.py)
    nlayers = int(rng.choice([3, 4, 9, 30, 70]))
    return dict(nlines=nlines, wnlow=wnlow, wnhigh=wnlow + width, wndelt=wndelt, wnosamp=osamp, nlayers=nlayers,
                solution=solution, toomuch=float(rng.choice([0.5, 5.0, 10.0, 50.0])),
                ethresh=float(rng.choice([1e-50, 1e-8, 1e-5, 1e-3])), ncia=int(rng.integers(0, 3)),
                seed=int(rng.integers(1, 10**6)), line_margin=float(rng.choice([0.0, 1.5])))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TRX_RANDOM_CASES", "48"))))
def test_random_problem_against_oracle(tmp_path, seed):
    kw = random_case(seed)
    d = str(tmp_path / "r")
    synth.make_case(d, **kw)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    ora = ol.OracleEngine(P.static)
    try:
        ref = ora.run(P.atm, P.opts, debug=DEBUG_KEYS)
    except EngineError as e:                     # e.g. fewer than three points for the modulation
        ref = e
    finally:
        ora.close()
    hip = Engine(P.static)
    try:
        for rep in range(3):                        # first run, then hinted runs
            if isinstance(ref, Exception):
                with pytest.raises(EngineError) as ei:
                    hip.run(P.atm, P.opts, debug=True)
                assert ei.value.code == ref.code, kw
                continue
            got = hip.run(P.atm, P.opts, debug=True)
            assert np.array_equal(got["last"], ref["last"]), (kw, rep)
            # optical depth: the arithmetic's 1e-9 plus the rounding steps of the reference's
            # absolute-radius parabola where they apply (tests/tolerances.py) -- nothing else
            noisy = assert_tau_close(P, got, ref, (kw, rep))
            # spectrum: 1e-8; rays whose optical depth carries parabola noise pass it on (in the
            # modulation of a 3-4-layer atmosphere R^2 - 2*integral cancels on top of it)
            assert rel_err(got["spectrum"][~noisy], ref["spectrum"][~noisy]) < 1e-8, (kw, rep)
            assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-7, (kw, rep)
            sw = got["computed"].astype(bool) & ref["computed"].astype(bool)
            assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-9, (kw, rep)
            assert rel_err(got["e_cs"], ref["e_cs"]) < 1e-12, (kw, rep)
    finally:
        hip.close()
