"""--saveext FILE: the reference keeps the molecular extinction of the layers a run computed "for a
possible next run" (savefile_extinct / restfile_extinct, extinction.c:62-137; read at the head of
tau(), tau.c:155-156, written at its end, :340-341).  Golden `saveext_transit` holds the file the
compiled reference wrote: magic "@E@S@", e[nlayer][nwn] doubles, one SHORT per layer (transit.h:129
defines _Bool as short)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from cases import GOLDEN, golden, rel_err
from transit_amd import build
from transit_amd.host import Problem

CASE = os.path.join(GOLDEN, "saveext_transit")


def ref_file():
    b = open(os.path.join(CASE, "ext.sav"), "rb").read()
    g = golden("saveext_transit")
    nr, nwn = g.problem.nlayer, g.problem.nwn
    assert b[:5] == b"@E@S@" and len(b) == 5 + 8 * nr * nwn + 2 * nr
    e = np.frombuffer(b[5:5 + 8 * nr * nwn]).reshape(nr, nwn)
    c = np.frombuffer(b[5 + 8 * nr * nwn:], dtype=np.int16)
    return e, c


def test_host_reads_and_writes_the_reference_s_file(tmp_path):
    e_ref, c_ref = ref_file()
    g = golden("saveext_transit")
    assert np.array_equal(c_ref != 0, g.swept)                    # the flags: the layers the lazy sweep reached
    assert rel_err(e_ref[g.swept], g.e[g.swept]) < 5e-9            # the rows: what mol_extion.dat prints at 10 digits
    d = tmp_path / "c"
    shutil.copytree(CASE, d)
    P = Problem.from_cfg(str(d / "case.cfg"))
    assert P.output_plan()["saveext"] == "ext.sav"
    e, c = P.saveext_read()
    assert np.array_equal(e, e_ref) and np.array_equal(c, (c_ref != 0).astype(np.uint8))
    os.remove(d / "ext.sav")
    assert P.saveext_read() is None and any("savefile" in m for m in P.messages())
    P.saveext_write(e, c)
    assert open(d / "ext.sav", "rb").read() == open(os.path.join(CASE, "ext.sav"), "rb").read()   # byte for byte
    (d / "ext.sav").write_bytes(b"@E@S@" + b"\0" * 100)            # another grid's file: refused, not followed
    assert P.saveext_read() is None


def test_oracle_restores_the_reference_s_file():
    """restfile_extinct + the lazy sweep: with the reference's own file restored, the layers it flags are
    never swept (the counters say so) and the spectrum is the reference's."""
    g = golden("saveext_transit")
    P = g.problem
    e_ref, c_ref = ref_file()
    ora = ol.OracleEngine(P.static)
    try:
        ora.restore_extinction(e_ref, (c_ref != 0).astype(np.uint8))
        got = ora.run(P.atm, P.opts, debug=("e", "computed"))
        assert ora.stats()["neval"] == 0                           # nothing was evaluated: every layer the rays need came from the file
        assert rel_err(got["spectrum"], g.spectrum) < 2e-8
        assert np.array_equal(got["e"], e_ref)
        e2 = e_ref.copy(); e2[c_ref != 0] *= 2.0                    # the restored rows are what the rays see
        ora.restore_extinction(e2, (c_ref != 0).astype(np.uint8))
        assert rel_err(ora.run(P.atm, P.opts)["spectrum"], g.spectrum) > 1e-3
        ora.restore_extinction(None)
        fresh = ora.run(P.atm, P.opts)
        assert ora.stats()["neval"] > 0 and rel_err(fresh["spectrum"], g.spectrum) < 2e-8
    finally:
        ora.close()


def _cli(work, *extra):
    exe = build.build_cli() or build.lib_path("transit_hip")
    for f in ("spectrum.dat", "toomuch.dat"):
        if os.path.exists(work / f):
            os.remove(work / f)
    return subprocess.run([exe, "-c", "case.cfg", *extra], cwd=work, capture_output=True, text=True, timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("ngpus", [1, 2])
def test_cli_writes_and_restores_the_file(tmp_path, ngpus):
    e_ref, c_ref = ref_file()
    ref_spec = ol.read_spectrum(os.path.join(CASE, "spectrum.dat"))[:, 1]
    work = tmp_path / "w"
    shutil.copytree(CASE, work)
    os.remove(work / "ext.sav")
    # no file yet: a note, the run, and the file of THIS run
    p = _cli(work, "--gpus", str(ngpus))
    assert p.returncode == 0, p.stderr
    assert "no extinction restored" in p.stderr
    b = open(work / "ext.sav", "rb").read()
    nr, nwn = e_ref.shape
    assert len(b) == 5 + 8 * nr * nwn + 2 * nr
    e = np.frombuffer(b[5:5 + 8 * nr * nwn]).reshape(nr, nwn); c = np.frombuffer(b[5 + 8 * nr * nwn:], dtype=np.int16)
    need = c_ref != 0
    assert np.all((c != 0)[need])                                  # at least the layers the reference's rays needed
    assert rel_err(e[need], e_ref[need]) < 1e-9
    assert np.all(e[c == 0] == 0)
    first = open(work / "spectrum.dat").read()
    # the reference's own file back in: its rows are used as they are -- doubled, the spectrum must move
    shutil.copy(os.path.join(CASE, "ext.sav"), work / "ext.sav")
    p = _cli(work, "--gpus", str(ngpus))
    assert p.returncode == 0 and "no extinction restored" not in p.stderr
    got = ol.read_spectrum(work / "spectrum.dat")[:, 1]
    assert rel_err(got, ref_spec) < 2e-8
    b2 = bytearray(open(os.path.join(CASE, "ext.sav"), "rb").read())
    e2 = np.frombuffer(bytes(b2[5:5 + 8 * nr * nwn])).reshape(nr, nwn).copy()
    e2[need] *= 2.0
    b2[5:5 + 8 * nr * nwn] = e2.tobytes()
    (work / "ext.sav").write_bytes(bytes(b2))
    p = _cli(work, "--gpus", str(ngpus))
    assert p.returncode == 0
    doubled = ol.read_spectrum(work / "spectrum.dat")[:, 1]
    assert rel_err(doubled, ref_spec) > 1e-3
    assert first != open(work / "spectrum.dat").read()


@pytest.mark.gpu
def test_restored_layers_are_not_swept_again():
    """trx_restore_extinction with every layer flagged: no line kernel runs (the counters of a counting run
    stay at zero) and the spectrum is the one of the run that produced the rows."""
    from transit_amd.engine import Engine
    P = golden("saveext_transit").problem
    eng = Engine(P.static)
    try:
        full = eng.run(P.atm, P.opts, debug=("e", "computed"))
        eng.restore_extinction(full["e"], np.ones(P.nlayer, dtype=np.uint8))
        P.opts.profile = 2
        try:
            again = eng.run(P.atm, P.opts)
            st = eng.stats()
        finally:
            P.opts.profile = 0
        assert st["neval"] == 0 and st["sum_bins"] == 0
        sw = full["computed"].astype(bool)
        assert sw.any()
        # (rows of layers the first run never swept are zero: below the deepest ray nothing reads them)
        assert rel_err(again["spectrum"], full["spectrum"]) < 1e-13
        eng.restore_extinction(None)
        P.opts.profile = 2
        try:
            eng.run(P.atm, P.opts)
            assert eng.stats()["neval"] > 0
        finally:
            P.opts.profile = 0
    finally:
        eng.close()
