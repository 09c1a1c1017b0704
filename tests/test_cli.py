"""`transit_hip`, the command-line drop-in for the reference's `transit`."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from cases import GOLDEN, rel_err
from transit_amd import build


def _run_cli(tmp_path, case):
    exe = build.build_cli() or build.lib_path("transit_hip")
    work = tmp_path / case
    shutil.copytree(os.path.join(GOLDEN, case), work)
    for f in ("spectrum.dat", "toomuch.dat", "intens.dat"):
        if os.path.exists(work / f):
            os.remove(work / f)
    p = subprocess.run([exe, "-c", "case.cfg"], cwd=work, capture_output=True, text=True, timeout=300)
    return work, p


def test_cli_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    work, p = _run_cli(tmp_path, "eclipse_small")
    assert p.returncode != 0
    assert "trx_create failed" in p.stderr
    assert not os.path.exists(work / "spectrum.dat")


def test_cli_rejects_unknown_option(tmp_path):
    exe = build.build_cli() or build.lib_path("transit_hip")
    p = subprocess.run([exe, "--nosuchoption", "1"], capture_output=True, text=True)
    assert p.returncode != 0 and "unknown" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["eclipse_small", "transit_small"])
def test_cli_writes_the_reference_spectrum_file(tmp_path, case):
    work, p = _run_cli(tmp_path, case)
    assert p.returncode == 0, p.stderr
    ref_txt = open(os.path.join(GOLDEN, case, "spectrum.dat")).read().split("\n")
    got_txt = open(work / "spectrum.dat").read().split("\n")
    assert got_txt[0] == ref_txt[0]                       # same header line
    assert len(got_txt) == len(ref_txt)
    got = ol.read_spectrum(work / "spectrum.dat")
    ref = ol.read_spectrum(os.path.join(GOLDEN, case, "spectrum.dat"))
    assert np.array_equal(got[:, 0], ref[:, 0])           # wavelength column: same text
    assert rel_err(got[:, 1], ref[:, 1]) < 2e-8
    # toomuch file: same layer index where the optical depth crossed the cut
    g = np.loadtxt(work / "toomuch.dat", comments="#", skiprows=2)
    r = np.loadtxt(os.path.join(GOLDEN, case, "toomuch.dat"), comments="#", skiprows=2)
    assert np.array_equal(g[:, 3], r[:, 3])
    assert rel_err(g[:, 1], r[:, 1]) < 1e-5


@pytest.mark.gpu
def test_cli_savefiles_dumps_match_the_reference_dumps(tmp_path):
    """`savefiles yes` (tau.c:180-190, 311-335): tau.dat, CIA.dat and mol_extion.dat in the
    reference's formats; the goldens hold the reference's own dumps of the same run."""
    case = "eclipse_small"
    exe = build.build_cli() or build.lib_path("transit_hip")
    work = tmp_path / case
    shutil.copytree(os.path.join(GOLDEN, case), work)
    for f in ("spectrum.dat", "toomuch.dat", "tau.dat", "CIA.dat", "mol_extion.dat"):
        if os.path.exists(work / f):
            os.remove(work / f)
    p = subprocess.run([exe, "-c", "case.cfg", "--savefiles", "yes"], cwd=work, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    for name, key in (("tau.dat", "wavenumber"), ("CIA.dat", "wavenumber"), ("mol_extion.dat", "radius")):
        ref_path = os.path.join(GOLDEN, case, name)
        got_lines = open(work / name).read().split("\n")
        ref_lines = open(ref_path).read().split("\n")
        assert got_lines[:4] == ref_lines[:4], name               # header block
        assert len(got_lines) == len(ref_lines), name
        gk, gv = ol.read_rows_dump(work / name, key)
        rk, rv = ol.read_rows_dump(ref_path, key)
        assert np.array_equal(gk, rk), name                       # row keys: same text
        assert np.array_equal(gv == 0, rv == 0), name             # incl. the layers the reference never swept
        assert rel_err(gv, rv) < 1e-8, name


@pytest.mark.gpu
def test_cli_outintens_matches_the_reference_file(tmp_path):
    """--outintens (printintens, eclipse.c:293-350): wavelength + one intensity column per angle."""
    work, p = _run_cli(tmp_path, "eclipse_small")
    assert p.returncode == 0, p.stderr
    ref_path = os.path.join(GOLDEN, "eclipse_small", "intens.dat")
    got_txt = open(work / "intens.dat").read().split("\n")
    ref_txt = open(ref_path).read().split("\n")
    assert got_txt[:2] == ref_txt[:2] and len(got_txt) == len(ref_txt)       # two header lines, same row count
    got, ref = np.loadtxt(work / "intens.dat", comments="#"), np.loadtxt(ref_path, comments="#")
    assert np.array_equal(got[:, 0], ref[:, 0])
    assert rel_err(got[:, 1:], ref[:, 1:]) < 2e-8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["qscale_eclipse", "dumps_transit"])
def test_cli_detail_and_sampling_files_match_the_reference(tmp_path, case):
    """detailext / detailtau / detailcia (detailout, tau.c:526-605) and outsample
    (makesample.c:744-770), written by the command the way the reference writes them."""
    exe = build.build_cli() or build.lib_path("transit_hip")
    work = tmp_path / case
    shutil.copytree(os.path.join(GOLDEN, case), work)
    outs = ("spectrum.dat", "toomuch.dat", "sample.dat", "detail_ext.dat", "detail_tau.dat", "detail_cia.dat",
            "tau.dat", "CIA.dat", "mol_extion.dat", "total_extion.dat", "cloud_extion.dat", "scatt_extion.dat")
    for f in outs:
        if os.path.exists(work / f):
            os.remove(work / f)
    p = subprocess.run([exe, "-c", "case.cfg"], cwd=work, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    ref = lambda f: os.path.join(GOLDEN, case, f)
    assert open(work / "sample.dat").read() == open(ref("sample.dat")).read()
    assert rel_err(ol.read_spectrum(work / "spectrum.dat")[:, 1], ol.read_spectrum(ref("spectrum.dat"))[:, 1]) < 2e-8
    from test_options import check_savefiles
    check_savefiles(str(work), case, 1e-8)                               # all six `savefiles` dumps
    for f in ("detail_tau.dat", "detail_ext.dat", "detail_cia.dat"):
        got, want = open(work / f).read().split("\n"), open(ref(f)).read().split("\n")
        assert got[0] == want[0] and len(got) == len(want), f            # picked wavenumbers, row count
        a, b = np.loadtxt(work / f, comments="#", ndmin=2), np.loadtxt(ref(f), comments="#", ndmin=2)
        assert np.array_equal(a[:, 0], b[:, 0]), f                       # radius / impact-parameter column
        if f == "detail_ext.dat":
            assert np.array_equal(a[:, 1:] == 0, b[:, 1:] == 0)          # the reference leaves unswept layers at zero
            assert rel_err(a[:, 1:], b[:, 1:]) < 2e-6, f
        elif f == "detail_tau.dat":
            assert np.array_equal(a[:, 1:] == 0, b[:, 1:] == 0)          # same toomuch cut
            assert rel_err(a[:, 1:], b[:, 1:]) < 2e-6, f
        else:
            # the reference prints 32-bit halves of its doubles (see trh_write_detail): the rows
            # showing the HIGH half (sign, exponent, 20 mantissa bits) are stable, compare those
            assert rel_err(a[1::2, 1:], b[1::2, 1:]) < 1e-5, f


@pytest.mark.gpu
@pytest.mark.parametrize("case,ngpus", [("eclipse_small", 2), ("transit_small", 3), ("coadd_thresh", 2)])
def test_cli_multi_gpu_shards_give_the_single_gpu_files(tmp_path, case, ngpus):
    """--gpus N: N shards of equal work, one handle and host thread each, the slices joined at the
    end (one ncclAllGather when there are N devices; in host memory when ranks share a device, as
    on the one-GPU test box).  Every output file must be the one-GPU run's."""
    exe = build.build_cli() or build.lib_path("transit_hip")
    outs = {}
    for n in (1, ngpus):
        work = tmp_path / ("%s_%d" % (case, n))
        shutil.copytree(os.path.join(GOLDEN, case), work)
        for f in ("spectrum.dat", "toomuch.dat", "intens.dat", "tau.dat", "CIA.dat", "mol_extion.dat", "total_extion.dat",
                  "cloud_extion.dat", "scatt_extion.dat"):
            if os.path.exists(work / f):
                os.remove(work / f)
        p = subprocess.run([exe, "-c", "case.cfg", "--savefiles", "yes", "--gpus", str(n)], cwd=work, capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        outs[n] = work
    # byte for byte, the extinction dumps too: the line sum of a bin is ordered the same way
    # whatever the shard, and the writers redo the reference's lazy-sweep zeros from the rays' depths
    for f in ("spectrum.dat", "toomuch.dat", "tau.dat", "CIA.dat", "mol_extion.dat", "total_extion.dat",
              "cloud_extion.dat", "scatt_extion.dat"):
        a, b = open(outs[1] / f).read(), open(outs[ngpus] / f).read()
        assert a == b, f
    got = ol.read_spectrum(outs[ngpus] / "spectrum.dat")
    ref = ol.read_spectrum(os.path.join(GOLDEN, case, "spectrum.dat"))
    assert rel_err(got[:, 1], ref[:, 1]) < 2e-8


def _run_ranks(tmp_path, case, ngpus, env=None):
    exe = build.build_cli() or build.lib_path("transit_hip")
    work = tmp_path / ("%s_%d" % (case, ngpus))
    shutil.copytree(os.path.join(GOLDEN, case), work)
    if os.path.exists(work / "spectrum.dat"):
        os.remove(work / "spectrum.dat")
    p = subprocess.run([exe, "-c", "case.cfg", "--gpus", str(ngpus)], cwd=work, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, **(env or {})))
    return work, p


def test_cli_ranks_without_gpu_stop_together(tmp_path):
    """--gpus 3 on a box without a device: every rank fails in its own phase, the job reports
    them all and ends -- no rank is left waiting in a collective (the timeout is the test)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    work, p = _run_ranks(tmp_path, "eclipse_small", 3)
    assert p.returncode != 0
    for k in range(3):
        assert "rank %d: trx_create failed" % k in p.stderr
    assert "no rank enters the next one" in p.stderr
    assert not os.path.exists(work / "spectrum.dat")


@pytest.mark.gpu
def test_cli_one_failing_rank_stops_the_job_before_the_gather(tmp_path):
    """One rank of three fails on its own (injected: TRANSIT_HIP_FAIL_RANK) while the others
    finish their spectra: the job must not enter the gather (with a communicator the other ranks
    would wait in ncclAllGather for ever), must name the rank and must not write a spectrum."""
    work, p = _run_ranks(tmp_path, "eclipse_small", 3, env={"TRANSIT_HIP_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert "rank 1: trx_run (TRANSIT_HIP_FAIL_RANK) failed" in p.stderr
    assert "rank 0:" not in p.stderr and "rank 2:" not in p.stderr
    assert "stopping after the spectrum phase" in p.stderr
    assert not os.path.exists(work / "spectrum.dat")
