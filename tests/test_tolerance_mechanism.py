"""CPU demonstration of the mechanism behind tests/tolerances.py: the reference's parabola in
absolute radius (interp_parab, pu/src/numerical.c:182-195, restated as trxo_parab3) returns its
own node's value only up to rounding steps of ulp(|m| (xr/dx)^2) -- a 1-ulp change of one input
moves the result by that much, ~1e5..1e9 ulps of the input for planetary radii."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from tolerances import NQ, parab_quantum


def parab(x, y, xr):
    lib = ol.oracle_library()
    xa, ya = (C.c_double * 3)(*x), (C.c_double * 3)(*y)
    return lib.trxo_parab3(xa, ya, float(xr))


@pytest.mark.parametrize("dx,min_amplification", [(1.0e7, 1e4), (1.0e5, 1e8)])
@pytest.mark.parametrize("node", [0, 1])        # the ray's bottom: x[0] (3+ layers left) or x[1] (two left)
def test_one_ulp_in_comes_back_as_a_rounding_step_of_the_cancelling_terms(dx, min_amplification, node):
    rng = np.random.default_rng(5)
    r0 = 7.0e9                                            # a Jupiter radius in cm
    worst_amp, worst_steps, moved = 0.0, 0.0, 0
    for _ in range(400):
        x = r0 + dx * np.arange(3) + rng.uniform(0, dx)
        y = 10.0 ** rng.uniform(-12, -6, size=3)          # extinction jumping between neighbouring layers
        q = parab_quantum(x[0], x[1], y[0], y[1], y[2], x[node])
        p = parab(x, y, x[node])
        # exact arithmetic returns the node's own value; doubles return it within a few steps
        assert abs(p - y[node]) <= NQ * q
        for k in range(3):
            y2 = y.copy()
            y2[k] = np.nextafter(y[k], np.inf)            # one ulp
            dp = abs(parab(x, y2, x[node]) - p)
            exact_change = (y2[k] - y[k]) if k == node else 0.0
            assert dp <= NQ * q + exact_change            # never more than the allowance ...
            if dp > exact_change:
                moved += 1
                worst_amp = max(worst_amp, dp / (y2[k] - y[k]))
                worst_steps = max(worst_steps, dp / q)
    assert moved > 100                                    # ... and it does happen, all the time
    assert worst_amp >= min_amplification                 # one ulp in, >= 1e4 (1e8) ulps out
    assert 0.05 <= worst_steps <= NQ                      # of the order of ONE rounding step: quantised noise,
                                                          # not a smooth error that shrinks with the perturbation


def test_smooth_extinction_has_no_noise_to_speak_of():
    """Where the extinction varies smoothly (second difference << value) the quantum is far
    below the arithmetic's own 1e-9: the allowance does not loosen anything there."""
    x = 7.0e9 + 1.0e7 * np.arange(3)
    y = 1e-8 * np.exp(-np.arange(3) * 1e-3)
    q = parab_quantum(x[0], x[1], y[0], y[1], y[2], x[0])
    assert q / y[0] < 1e-9
