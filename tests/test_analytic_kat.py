"""The analytic known-answer tests the reference keeps for its ray solutions
(transit/test/test_slantpath.c, tolerance 1e-4, :62), re-expressed:

  * optical depth of a slant ray through a sphere whose extinction is constant, grows outwards
    (alpha r) or inwards (alpha (rm - r))                 (:177-198, analytic values :180-197)
  * modulation for optical depths that are constant, grow or fall with the impact parameter
                                                           (:231-307, analytic values :236-296)

on the CPU restatement (its totaltau1 / modulation1) and, end to end through the C ABI, on the HIP
path: a line-free atmosphere whose extinction comes from the grey-cloud and the Lecavelier
scattering models (extinction.c:587-693) gives a sphere with constant or r-proportional
extinction, whose optical depths AND modulation have closed forms.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from cases import golden
from transit_amd import _abi

TOL = 1e-4            # test_slantpath.c:62 (maxerr)
DP = C.POINTER(C.c_double)


def sphere(alpha, rm, nrad):
    """test_slantpath.c:110-124 (tau_dens) and :67-78 (calcex)."""
    rad = (np.arange(nrad) + 1.0) * (rm / nrad)
    ex = np.ones(nrad) if alpha == 0 else (-alpha * (rm - rad) if alpha < 0 else alpha * rad)
    return rad, ex


def tau_analytic(alpha, rm, ip):
    """test_slantpath.c:177-198."""
    rat = rm / ip
    if alpha == 0:
        return 2 * np.sqrt(rm * rm - ip * ip)
    root = rm * ip * np.sqrt(rat * rat - 1)
    log = ip * ip * np.log(np.sqrt(rat * rat - 1) + rat)
    return alpha * (root + log) if alpha > 0 else -alpha * (root - log)


@pytest.mark.parametrize("alpha", [1.0, -1.0, 0.0])
@pytest.mark.parametrize("rm", [10.0, 100.0, 1000.0])
def test_oracle_optical_depth_through_a_sphere(alpha, rm):
    lib = ol.oracle_library()
    # 1000 layers: the reference's tolerance.  100 layers: the quadrature itself (trapezoid on the
    # first interval when the point count is even, numerical.c:500-525) is only good to ~2e-3 for
    # a ray starting three quarters out -- a property of the algorithm, which the restatement
    # shares with the compiled reference to 1e-9 (tests/test_oracle_golden.py)
    for nrad, tol in ((1000, TOL), (100, 3e-3)):
        rad, ex = sphere(alpha, rm, nrad)
        for frac in (0.1, 0.5, 0.75, 0.9):
            ip = frac * rm
            got = lib.trxo_tau_slant(rad.ctypes.data_as(DP), nrad, ip, ex.copy().ctypes.data_as(DP))
            assert abs(1 - got / tau_analytic(alpha, rm, ip)) < tol, (alpha, rm, nrad, frac)


def mod_case(kind, prm, star, ipmax, first, nip, toomuch):
    """test_slantpath.c:231-307 (mod_ctau / mod_itau / mod_dtau): (tau[nip], expected)."""
    rath, ratl = ipmax / star, first * ipmax / star
    delt = (1 - first) / (nip - 1)
    k = np.arange(nip)
    if kind == "constant":
        tau = np.full(nip, prm)
        res = -np.exp(-prm) * (rath * rath - ratl * ratl)
    elif kind == "increasing":
        tau = prm * ipmax * (1 - k * delt)
        res = -2 * (np.exp(-prm * ipmax * first) * (first * ipmax + 1 / prm) - np.exp(-prm * ipmax) * (ipmax + 1 / prm)) / star / star / prm
    else:
        tau = prm * ipmax * k * delt
        res = -2 * ((ipmax - 1 / prm) - np.exp(-prm * ipmax * (1 - first)) * (ipmax * first - 1 / prm)) / star / star / prm
    return tau, res - np.exp(-toomuch) * ratl * ratl + rath * rath


@pytest.mark.parametrize("kind", ["constant", "increasing", "decreasing"])
def test_oracle_modulation_for_prescribed_optical_depths(kind):
    lib = ol.oracle_library()
    toomuch, prm = 30.0, 0.01                                   # test_slantpath.c:499, 506-511
    for star in (10.0, 100.0, 1000.0):
        for ipmax in (10.0, 100.0, 1000.0):
            if ipmax > star:                                    # :425-426
                continue
            for first in (0.9, 0.75, 0.5, 0.1):
                for nip in (100, 1000):
                    tau, want = mod_case(kind, prm, star, ipmax, first, nip, toomuch)
                    delt = ipmax * (1 - first) / (nip - 1)
                    ip = first * ipmax + (nip - 1 - np.arange(nip)) * delt        # :352-353, top first
                    got = lib.trxo_modulation(tau.ctypes.data_as(DP), nip - 1, toomuch, ip.ctypes.data_as(DP), nip, 1.0, star, 1)
                    assert abs(1 - got / want) < TOL, (kind, star, ipmax, first, nip)


# ---- the same physics end to end through the C ABI on the GPU ------------------------------
def sphere_problem(nlay, rm, law, kappa):
    """trx_static/atm/opts of a line-free, CIA-free atmosphere of nlay layers up to radius rm
    whose only extinction is kappa (law 'const': grey cloud deck, extinction.c:662-666) or
    kappa*r (law 'out': Lecavelier scattering with p/T = r, extinction.c:605-610)."""
    P = golden("transit_small").problem
    st = _abi.TrxStatic.from_buffer_copy(P.static)
    st.nlines = 0; st.ncia = 0
    opts = _abi.TrxOpts.from_buffer_copy(P.opts)
    opts.solution = 1; opts.toomuch = 1e300; opts.modlevel = 1; opts.eager = 1
    rad = (np.arange(nlay) + 1.0) * (rm / nlay)
    nm, ni = st.nmol, st.niso
    keep = dict(rad=rad, temp=np.ones(nlay), press=np.ones(nlay), dens=np.full(nm * nlay, 1e-12), ab=np.full(nm * nlay, 1.0 / nm),
                z=np.ones(max(ni, 1) * nlay))
    wn0 = st.wn_i
    if law == "const":
        opts.cloud_flag = 1; opts.cloud_ext = kappa; opts.cloud_top = -5.0; opts.cloud_bot = 5.0; opts.scat_flag = 0
    else:
        opts.cloud_flag = 0; opts.scat_flag = 1
        e0h2 = 4.911e-23                                         # E0H2 (constants_tr.h), extinction.c:608
        keep["press"] = rad.copy()                                # p/T = r
        opts.scat_logext = np.log10(kappa / (e0h2 * wn0 ** 4))
    atm = _abi.TrxAtm(nlay, 1.0, keep["rad"].ctypes.data_as(DP), keep["temp"].ctypes.data_as(DP), keep["press"].ctypes.data_as(DP),
                      keep["dens"].ctypes.data_as(DP), keep["ab"].ctypes.data_as(DP), keep["z"].ctypes.data_as(DP))
    return st, atm, opts, keep


@pytest.mark.gpu
@pytest.mark.parametrize("law", ["const", "out"])
@pytest.mark.parametrize("nlay,tol", [(100, 3e-3), (1000, TOL)])
def test_gpu_optical_depth_through_a_sphere(law, nlay, tol):
    from transit_amd.engine import Engine
    rm, kappa = 1000.0, 3e-3
    st, atm, opts, keep = sphere_problem(nlay, rm, law, kappa)
    eng = Engine(st)
    out = eng.run(atm, opts, debug=("tau", "last"))
    eng.close()
    tau = out["tau"][0]                                           # first wavenumber; heights top first
    for frac in (0.1, 0.5, 0.75, 0.9):
        k = int(round(frac * nlay)) - 1                           # layer whose radius is frac*rm
        b = keep["rad"][k]
        want = kappa * tau_analytic(0.0 if law == "const" else 1.0, rm, b)
        got = tau[nlay - 1 - k]
        assert abs(1 - got / want) < tol, (law, nlay, frac, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("transparent", [0, 1])
def test_gpu_modulation_of_a_constant_extinction_sphere(transparent):
    """tau(b) = 2 kappa sqrt(R^2 - b^2) in closed form, and with u = sqrt(R^2 - b^2)
    int exp(-tau) b db = [1 - exp(-2 kappa U)(1 + 2 kappa U)] / (4 kappa^2), U = u(b_min):
    optical depth AND modulation kernels against an exact answer."""
    from transit_amd.engine import Engine
    nlay, rm, kappa = 1000, 1000.0, 2e-3
    st, atm, opts, keep = sphere_problem(nlay, rm, "const", kappa)
    opts.transparent = transparent
    opts.starrad_cm = 5000.0
    eng = Engine(st)
    out = eng.run(atm, opts, debug=("tau", "last"))
    eng.close()
    bmin = keep["rad"][0]
    U = np.sqrt(rm * rm - bmin * bmin)
    integral = (1 - np.exp(-2 * kappa * U) * (1 + 2 * kappa * U)) / (4 * kappa * kappa)
    want = rm * rm - 2 * integral
    if transparent:
        want -= np.exp(-max(out["tau"][0][nlay - 1], opts.toomuch)) * bmin * bmin
    want /= opts.starrad_cm ** 2
    assert abs(1 - out["spectrum"][0] / want) < TOL
