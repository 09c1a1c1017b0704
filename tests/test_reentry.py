"""Library re-entry (the BART path): transit_init once, then run_transit(T, abundances)
repeatedly (transit.c:118-122 -> reloadatm + radpress + makeradsample, readatm.c:722-865).
Golden values come from the compiled reference's own library API at full double
precision (oracle/ref_reentry_main.c drives it)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from cases import GOLDEN, rel_err
from transit_amd.host import HostError, Problem

CASE = os.path.join(GOLDEN, "reentry")
INPUTS = np.loadtxt(os.path.join(CASE, "reentry_inputs.txt"))
EXPECT = [np.loadtxt(os.path.join(CASE, "reentry_out%d.dat" % (k + 1))) for k in range(len(INPUTS))]
CASE_T = os.path.join(GOLDEN, "reentry_transit")
EXPECT_T = [np.loadtxt(os.path.join(CASE_T, "reentry_out%d.dat" % (k + 1))) for k in range(len(INPUTS))]


def run_sequence(engine_cls, case=CASE):
    P = Problem.from_cfg(os.path.join(case, "case.cfg"))
    eng = engine_cls(P.static)              # static data (lines, table) made once
    outs = []
    for vec in INPUTS:
        P.reload_atm(vec)
        outs.append(eng.run(P.atm, P.opts)["spectrum"])
    eng.close()
    return outs


def test_oracle_follows_the_reference_through_three_reloads():
    for got, ref in zip(run_sequence(ol.OracleEngine), EXPECT):
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 1e-9


def test_hydrostatic_radii_are_the_reference_s_bit_for_bit():
    """radpress (readatm.c:787-865) + makeradsample: the radii after every reload, against the
    reference's own rads.v dumped at %.17g by oracle/ref_reentry_main.c."""
    for case in (CASE, CASE_T):
        P = Problem.from_cfg(os.path.join(case, "case.cfg"))
        for k, vec in enumerate(INPUTS):
            P.reload_atm(vec)
            ref = np.loadtxt(os.path.join(case, "reentry_out%d_radii.dat" % (k + 1)))
            assert np.array_equal(P.layer_arrays()["radius"], ref), (case, k)


def test_transit_geometry_reloads_follow_the_reference():
    """Transmission geometry after radpress().  The impact parameter the reference hands to its
    ray solution is h*hfct/rfct (tau.c:274), which its object code -- built with -ffast-math --
    evaluates as (h*hfct)*(1/rfct).  With that arithmetic the restatement follows the reference
    to round-off.  (Evaluated as a true division, as the restatement did at first, two of the
    twenty hydrostatic radii come out an ulp BELOW the layer radius, the slant-path bracket
    search, slantpath.c:36, starts a layer lower for them and the Simpson pairing of the whole ray
    shifts: the 2e-3 this test used to allow was the restatement's, not the reference's.)"""
    for got, ref in zip(run_sequence(ol.OracleEngine, CASE_T), EXPECT_T):
        assert rel_err(got, ref) < 1e-9


def test_the_form_of_the_impact_parameter_decides_the_brackets():
    """The mechanism, isolated on the reference's own radii: the reciprocal form lands on the
    radius or an ulp above it (same bracket, closest approach an ulp higher: a 1e-10 effect);
    the division form lands an ulp below for some hydrostatic radii (bracket one layer lower:
    a 1e-3 effect on those rays).  File radii (short decimals) survive both forms' brackets."""
    fct = 1e5
    for k in (1, 2, 3):
        ref = np.loadtxt(os.path.join(CASE_T, "reentry_out%d_radii.dat" % k))
        recip, div = (ref * fct) * (1.0 / fct), (ref * fct) / fct
        assert np.count_nonzero(recip < ref) == 0 and np.count_nonzero(recip > ref) > 0
        if k == 1:
            assert np.count_nonzero(div < ref) > 0
    for case in ("transit_small", "cloud_scatter", "transit_modm1", "midres_os4", "dumps_transit"):
        g = Problem.from_cfg(os.path.join(GOLDEN, case, "case.cfg")).layer_arrays()["radius"]
        assert np.count_nonzero((g * fct) * (1.0 / fct) < g) == 0 and np.array_equal((g * fct) / fct, g)


def test_reload_needs_reference_level_options(tmp_path):
    import shutil
    d = tmp_path / "c"
    shutil.copytree(CASE, d)
    cfg = (d / "case.cfg").read_text().replace("gsurf 1000.0\n", "")
    (d / "case.cfg").write_text(cfg)
    P = Problem.from_cfg(str(d / "case.cfg"))
    with pytest.raises(HostError):          # readatm.c:760-767: gsurf, refpress, refradius are mandatory
        P.reload_atm(INPUTS[0])
    with pytest.raises(HostError):
        Problem.from_cfg(os.path.join(CASE, "case.cfg")).reload_atm(INPUTS[0][:-1])


@pytest.mark.gpu
def test_gpu_follows_the_reference_through_three_reloads():
    from transit_amd.engine import Engine
    gpu = run_sequence(Engine)
    cpu = run_sequence(ol.OracleEngine)
    for got, ora, ref in zip(gpu, cpu, EXPECT):
        assert rel_err(got, ref) < 1e-9
        assert rel_err(got, ora) < 1e-10
    # transmission geometry: same impact parameters, brackets and point sets as the reference
    for got, ora, ref in zip(run_sequence(Engine, CASE_T), run_sequence(ol.OracleEngine, CASE_T), EXPECT_T):
        assert rel_err(got, ref) < 1e-8
        assert rel_err(got, ora) < 1e-8


@pytest.mark.gpu
def test_swig_module_lookalike_follows_the_reference():
    """transit_amd.transit_module = the reference's `transit_module` (transit.i:97-105): same
    function names and signatures, driven the way BART drives it."""
    import transit_amd.transit_module as trm
    old = os.getcwd()
    os.chdir(CASE)
    try:
        argv = ["transit", "-c", "case.cfg"]
        trm.transit_init(len(argv), argv)
        n = trm.get_no_samples()
        wn = trm.get_waveno_arr(n)
        assert n == EXPECT[0].size and wn.shape == (n,) and np.all(np.diff(wn) > 0)
        for vec, ref in zip(INPUTS, EXPECT):
            out = trm.run_transit(vec, n)
            assert out.shape == (n,) and rel_err(out, ref) < 1e-9
        trm.free_memory()
        with pytest.raises(RuntimeError):
            trm.get_no_samples()
    finally:
        os.chdir(old)
        trm.free_memory()


# ---- the setters between runs: set_radius / set_cloudtop / set_scattering (transit.c:97-116) ----
CASE_S = os.path.join(GOLDEN, "reentry_set")
CASE_ST = os.path.join(GOLDEN, "reentry_set_transit")


def setter_script(case):
    """reentry_inputs.txt of the setter goldens: ("radius", r) / ("cloudtop", c) / ("scattering", flag, x)
    between ("run", vector) entries, in file order (oracle/ref_reentry_main.c reads the same file)."""
    out = []
    for ln in open(os.path.join(case, "reentry_inputs.txt")):
        w = ln.split()
        if not w:
            continue
        if w[0] in ("radius", "cloudtop", "scattering"):
            out.append((w[0],) + tuple(float(x) for x in w[1:]))
        else:
            out.append(("run", np.array([float(x) for x in w])))
    return out


def run_setter_sequence(engine_cls, case):
    """Spectra and radii of the setter script through trh_set_* + trh_reload_atm + the engine."""
    P = Problem.from_cfg(os.path.join(case, "case.cfg"))
    eng = engine_cls(P.static)
    outs, radii = [], []
    for step in setter_script(case):
        if step[0] == "radius":
            P.set_radius(step[1])
        elif step[0] == "cloudtop":
            P.set_cloudtop(step[1])
        elif step[0] == "scattering":
            P.set_scattering(int(step[1]), step[2])
        else:
            P.reload_atm(step[1])
            radii.append(P.layer_arrays()["radius"].copy())
            outs.append(eng.run(P.atm, P.opts)["spectrum"])
    eng.close()
    return outs, radii


def expected(case):
    n = sum(1 for s in setter_script(case) if s[0] == "run")
    return ([np.loadtxt(os.path.join(case, "reentry_out%d.dat" % (k + 1))) for k in range(n)],
            [np.loadtxt(os.path.join(case, "reentry_out%d_radii.dat" % (k + 1))) for k in range(n)])


def test_setter_script_uses_all_three_setters_and_they_matter():
    for case in (CASE_S, CASE_ST):
        kinds = [s[0] for s in setter_script(case)]
        assert kinds == ["run", "radius", "run", "cloudtop", "run", "scattering", "run"]
        spec, _ = expected(case)
        # the last run repeats the first atmosphere: what differs is the setters' doing
        assert rel_err(spec[3], spec[0]) > 1e-2 and rel_err(spec[1], spec[0]) > 1e-2


@pytest.mark.parametrize("case,tol", [(CASE_S, 1e-9), (CASE_ST, 1e-9)])
def test_oracle_follows_the_reference_through_the_setters(case, tol):
    got, radii = run_setter_sequence(ol.OracleEngine, case)
    spec, rad = expected(case)
    for k in range(len(spec)):
        assert np.array_equal(radii[k], rad[k]), (case, k)          # set_radius moves the hydrostatic radii: bit for bit
        assert rel_err(got[k], spec[k]) < tol, (case, k)


@pytest.mark.gpu
@pytest.mark.parametrize("case,tol", [(CASE_S, 1e-9), (CASE_ST, 1e-8)])
def test_gpu_follows_the_reference_through_the_setters(case, tol):
    from transit_amd.engine import Engine
    got, radii = run_setter_sequence(Engine, case)
    spec, rad = expected(case)
    for k in range(len(spec)):
        assert np.array_equal(radii[k], rad[k]), (case, k)
        assert rel_err(got[k], spec[k]) < tol, (case, k)


@pytest.mark.gpu
@pytest.mark.parametrize("case,tol", [(CASE_S, 1e-9), (CASE_ST, 1e-8)])
def test_swig_module_lookalike_setters_follow_the_reference(case, tol):
    """set_radius / set_cloudtop / set_scattering of transit_amd.transit_module, driven as BART does."""
    import transit_amd.transit_module as trm
    old = os.getcwd()
    os.chdir(case)
    try:
        argv = ["transit", "-c", "case.cfg"]
        trm.transit_init(len(argv), argv)
        n = trm.get_no_samples()
        spec, _ = expected(case)
        k = 0
        for step in setter_script(case):
            if step[0] == "radius":
                trm.set_radius(step[1])
            elif step[0] == "cloudtop":
                trm.set_cloudtop(step[1])
            elif step[0] == "scattering":
                trm.set_scattering(int(step[1]), step[2])
            else:
                assert rel_err(trm.run_transit(step[1], n), spec[k]) < tol, (case, k)
                k += 1
    finally:
        os.chdir(old)
        trm.free_memory()
