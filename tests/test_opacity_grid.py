"""Opacity-grid mode (--opacityfile): build (calcopacity, opacity.c:282-427), file
format (:405-421), reuse (readopacity :433-503, interpolmolext extinction.c:535-581),
against the grid file and the spectrum the compiled reference produced."""
import os
import shutil

import numpy as np
import pytest

import oracle_lib as ol
from cases import GOLDEN, rel_err
from transit_amd.host import Problem

CASE = os.path.join(GOLDEN, "opacity_grid")


def read_grid(path):
    with open(path, "rb") as f:
        nmol, ntemp, nlayer, nwave = np.fromfile(f, dtype=np.int64, count=4)
        ids = np.fromfile(f, dtype=np.int32, count=nmol)
        temp = np.fromfile(f, dtype=np.float64, count=ntemp)
        press = np.fromfile(f, dtype=np.float64, count=nlayer)
        wns = np.fromfile(f, dtype=np.float64, count=nwave)
        o = np.fromfile(f, dtype=np.float64).reshape(nlayer, ntemp, nmol, nwave)
    return ids, temp, press, wns, o


def workdir(tmp_path, with_file):
    d = tmp_path / "og"
    shutil.copytree(CASE, d)
    if with_file:
        shutil.copy(d / "opacity_ref.dat", d / "opac.dat")
    return str(d)


REF = read_grid(os.path.join(CASE, "opacity_ref.dat"))
REF_SPEC = ol.read_spectrum(os.path.join(CASE, "spectrum.dat"))[:, 1]


def check_build(engine_cls, d, tol):
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.needs_opacity_build and not P.static.ogrid
    eng = engine_cls(P.static)
    o = eng.build_opacity_grid(P)
    eng.close()
    assert not P.needs_opacity_build and P.static.ogrid
    ids, temp, press, wns, og = read_grid(os.path.join(d, "opac.dat"))
    rids, rtemp, rpress, rwns, rog = REF
    assert np.array_equal(ids, rids) and np.array_equal(temp, rtemp) and np.array_equal(wns, rwns)
    assert rel_err(press, rpress) < 1e-14
    assert og.shape == rog.shape == (16, 5, 2, 41)
    assert np.array_equal(og.reshape(o.shape), o)
    assert np.array_equal(og == 0, rog == 0)              # same thresholded / untouched bins
    assert rel_err(og, rog) < tol
    return P


def check_run(engine_cls, P, tol):
    eng = engine_cls(P.static)
    out = eng.run(P.atm, P.opts, debug=True)
    eng.close()
    assert rel_err(out["spectrum"], REF_SPEC) < tol
    return out


def test_oracle_builds_the_reference_grid_and_spectrum(tmp_path):
    P = check_build(ol.OracleEngine, workdir(tmp_path, False), 1e-9)
    check_run(ol.OracleEngine, P, 2e-8)


def test_oracle_reads_an_existing_grid(tmp_path):
    d = workdir(tmp_path, True)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert not P.needs_opacity_build and P.static.ogrid and P.static.nlines == 0    # TLI not read
    check_run(ol.OracleEngine, P, 2e-8)


@pytest.mark.gpu
def test_gpu_builds_the_reference_grid_and_spectrum(tmp_path):
    from transit_amd.engine import Engine
    d = workdir(tmp_path, False)
    P = check_build(Engine, d, 1e-9)
    got = check_run(Engine, P, 2e-8)
    ref = check_run(ol.OracleEngine, P, 2e-8)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert rel_err(got["e"][ref["computed"].astype(bool)], ref["e"][ref["computed"].astype(bool)]) < 1e-12


@pytest.mark.gpu
def test_gpu_per_molecule_sweep_matches_oracle(tmp_path):
    from transit_amd.engine import Engine
    P = Problem.from_cfg(os.path.join(workdir(tmp_path, False), "case.cfg"))
    nv, t, dn, z, nslot, sl = P.grid_request()
    hip, ora = Engine(P.static), ol.OracleEngine(P.static)
    a = hip.sweep_permol(nv, t, dn, z, P.opts.ethresh, nslot, sl)
    b = ora.sweep_permol(nv, t, dn, z, P.opts.ethresh, nslot, sl)
    hip.close(); ora.close()
    assert a.shape == (80, 2, 41) and (b != 0).any()
    assert np.array_equal(a == 0, b == 0)
    assert rel_err(a, b) < 1e-10


@pytest.mark.gpu
def test_gpu_rejects_a_layer_outside_the_grid(tmp_path):
    from transit_amd import _abi
    from transit_amd.engine import Engine, EngineError
    P = Problem.from_cfg(os.path.join(workdir(tmp_path, True), "case.cfg"))
    eng = Engine(P.static)
    hot = P.layer_arrays()["temp"].copy(); hot[3] = 1950.0
    a = _abi.TrxAtm.from_buffer_copy(P.atm); a.temp = hot.ctypes.data_as(_abi.c_double_p)
    with pytest.raises(EngineError) as ei:
        eng.run(a, P.opts)
    assert ei.value.code == -5
    eng.close()
