"""No option of the reference's table (argum.c:112-320) may be accepted and then dropped.

For EVERY entry of the host side's table this file holds one probe: a value to set and the
observable it must change -- a field of the plain structs handed to the engine, the wavenumber
grid, the list of files a run writes, the recorded messages, or an error.  A table entry
without a probe fails the test, so a new option cannot be added without saying what it does.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from cases import GOLDEN, golden, rel_err
from transit_amd import _abi, build
from transit_amd.host import HostError, Problem, option_table

CASE = "qscale_eclipse"          # has refpress-free cfg with every output option of the reference


def digest(P):
    """Everything the engine and the writers get to see, as one comparable dict."""
    st, a, o = P.static, P.atm, P.opts
    d = {}
    for name, _t in _abi.TrxStatic._fields_:
        v = getattr(st, name)
        if not isinstance(v, (int, float)):
            continue
        d["st." + name] = v
    for name, _t in _abi.TrxOpts._fields_:
        v = getattr(o, name)
        if isinstance(v, (int, float)):
            d["o." + name] = v
    d["o.angles"] = tuple(o.angles_deg[i] for i in range(o.nangles))
    d["a.nlayer"], d["a.rad_fct"] = a.nlayer, a.rad_fct
    for k, v in P.layer_arrays().items():
        d["a." + k] = v.tobytes()
    n = st.nlines
    d["st.lines"] = np.ctypeslib.as_array(st.wl_um, shape=(n,)).tobytes() if n else b""
    if st.ncia:
        d["st.cia0"] = (st.cia[0].nwave, st.cia[0].ntemp)
    d["plan"] = tuple(sorted(P.output_plan().items()))
    d["verb"] = P.option("verb")
    d["msgs"] = tuple(P.messages())
    d["reload"] = tuple(P.option(k) for k in ("refpress", "refradius", "gsurf"))
    d["grid"] = tuple(P.option(k) for k in ("opacityfile", "tlow", "thigh", "tempdelt", "justOpacity"))
    return d


def base_args():
    return ["-c", "case.cfg"]


# name -> (extra argv, check(base_digest, new_digest or exception))
def changed(*keys):
    def chk(b, n):
        assert not isinstance(n, Exception), n
        for k in keys:
            assert b.get(k) != n.get(k), "option did not change %s" % k
    return chk


def fails(code=None):
    def chk(b, n):
        assert isinstance(n, HostError), "expected an error, got a problem"
        if code is not None:
            assert n.code == code
    return chk


def noted(word, level="I"):
    def chk(b, n):
        assert not isinstance(n, Exception), n
        assert any(m.startswith(level + ":") and word in m for m in n["msgs"]), n["msgs"]
        b2, n2 = dict(b), dict(n)
        b2.pop("msgs"), n2.pop("msgs")
        assert b2 == n2                    # ... and nothing else: the numbers cannot change
    return chk


PROBES = {
    "version": (["--version"], fails(1)),                     # served: text on stdout, nothing to run
    "help": (["--help"], fails(1)),
    "quiet": (["--quiet"], changed("verb")),
    "verb": (["--verb", "4"], changed("verb")),
    "config_file": (["-c", "case.cfg", "--config_file", "more.cfg"], changed("o.toomuch")),    # more.cfg: toomuch 7
    "atm": (["--atm", "other.atm"], changed("a.temp")),
    "linedb": (["--linedb", "other.tli"], changed("st.lines")),
    "outtoomuch": (["--outtoomuch", "tm2.dat"], changed("plan")),
    "outsample": (["--outsample", "s2.dat"], changed("plan")),
    "outspec": (["--outspec", "sp2.dat"], changed("plan")),
    "outintens": (["--outintens", "i2.dat"], changed("plan")),
    "molfile": (["--molfile", "mol2.dat"], changed("a.density")),
    "savefiles": (["--savefiles", "no"], changed("plan")),
    "raddelt": (["--raddelt", "400"], changed("a.nlayer")),
    "radlow": (["--raddelt", "400", "--radlow", "93000"], changed("a.radius")),
    "radhigh": (["--raddelt", "400", "--radhigh", "99000"], changed("a.radius")),
    "radfct": (["--raddelt", "400", "--radfct", "1.1e5"], changed("a.rad_fct")),
    "allowq": (["--allowq", "-1"], changed("msgs")),          # every layer now warns about its abundance sum
    "refpress": (["--refpress", "0.1"], changed("reload")),
    "refradius": (["--refradius", "95000"], changed("reload")),
    "gsurf": (["--gsurf", "1000"], changed("reload")),
    "qmol": (["--qmol", "CO CO2"], changed("a.density")),
    "qscale": (["--qscale", "0.1 0.2"], changed("a.density")),
    "wllow": (["--wnhigh", "0", "--wllow", "3.95"], changed("st.nwn")),
    "wlhigh": (["--wnlow", "0", "--wlhigh", "3.99"], changed("st.wn_i")),
    "wlfct": (["--wnlow", "0", "--wlhigh", "39900", "--wlfct", "1e-8"], changed("st.wn_i")),
    "wnlow": (["--wnlow", "2510"], changed("st.wn_i")),
    "wnhigh": (["--wnhigh", "2530"], changed("st.nwn")),
    "wndelt": (["--wndelt", "0.5"], changed("st.wn_d")),
    "wnosamp": (["--wnosamp", "1080"], changed("st.osamp")),
    "wnfct": (["--wnfct", "0"], fails(-1)),                   # makesample.c:318-323: must be positive with wnlow
    "ndop": (["--ndop", "40"], changed("st.ndop")),
    "nlor": (["--nlor", "40"], changed("st.nlor")),
    "dmin": (["--dmin", "2e-3"], changed("st.dmin")),
    "dmax": (["--dmax", "0.3"], changed("st.dmax")),
    "lmin": (["--lmin", "2e-4"], changed("st.lmin")),
    "lmax": (["--lmax", "8"], changed("st.lmax")),
    "nwidth": (["--nwidth", "10"], changed("st.timesalpha")),
    "ethreshold": (["--ethreshold", "1e-6"], changed("o.ethresh")),
    "cloud": (["--cloud", "ext,1e-5,-3,0"], changed("o.cloud_flag", "o.cloud_ext")),
    "cloudtop": (["--cloudtop", "-2"], changed("o.cloud_flag", "o.cloud_top")),
    "scattering": (["--scattering", "polar"], changed("o.scat_flag")),
    "detailext": (["--detailext", "de2.dat:2505"], changed("plan")),
    "detailcia": (["--detailcia", "dc2.dat:2505"], changed("plan")),
    "csfile": (["--csfile", "cia2.dat"], changed("st.cia0")),
    "saveext": (["--saveext", "ext.sav"], changed("plan")),
    "opacityfile": (["--opacityfile", "opa.dat", "--tlow", "1100", "--thigh", "1900", "--tempdelt", "200"],
                    changed("plan", "grid")),
    "tlow": (["--tlow", "600"], changed("grid")),
    "thigh": (["--thigh", "2500"], changed("grid")),
    "tempdelt": (["--tempdelt", "50"], changed("grid")),
    "justOpacity": (["--justOpacity"], changed("grid")),
    "shareOpacity": (["--shareOpacity"], noted("shareOpacity", "W")),
    "solution": (["--solution", "transit"], changed("o.solution", "o.nangles")),
    "toomuch": (["--toomuch", "5"], changed("o.toomuch")),
    "taulevel": (["--taulevel", "2"], fails(-6)),
    "modlevel": (["--modlevel", "-1"], changed("o.modlevel")),
    "detailtau": (["--detailtau", "dt2.dat:2505"], changed("plan")),
    "starrad": (["--starrad", "0.9"], changed("o.starrad_cm")),
    "gorbpar": (["--gorbpar", "1,0,0,0,0,0"], noted("gorbpar")),
    "gorbparfct": (["--gorbparfct", "1,1,1,1,1,1"], noted("gorbparfct")),
    "transparent": (["--transparent"], changed("o.transparent")),
    "raygrid": (["--raygrid", "0 30 60"], changed("o.nangles", "o.angles")),
}


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("opts") / CASE
    shutil.copytree(os.path.join(GOLDEN, CASE), d)
    (d / "more.cfg").write_text("toomuch 7\n")
    # a second atmosphere (hotter), line list, molecule table (heavier CH4) and CIA table
    atm = (d / "case.atm").read_text().split("\n")
    k = next(i for i, l in enumerate(atm) if l.startswith("#Radius")) + 1
    rows = []
    for l in atm[k:]:
        w = l.split()
        if len(w) > 3:
            w[2] = "%.3f" % (float(w[2]) + 25.0)
        rows.append("  ".join(w))
    (d / "other.atm").write_text("\n".join(atm[:k] + rows) + "\n")
    shutil.copy(os.path.join(GOLDEN, "reentry", "case.tli"), d / "other.tli")
    (d / "mol2.dat").write_text((d / "molecules.dat").read_text().replace("16.0425", "17.0425"))
    cia = (d / "cia_h2h2.dat").read_text().split("\n")
    data = [i for i, l in enumerate(cia) if l and l[0] not in "#it@"]
    extra = cia[data[-1]].split()
    extra[0] = "%g" % (float(extra[0]) + 500.0)
    (d / "cia2.dat").write_text("\n".join(cia[:data[-1] + 1] + ["  ".join(extra)] + cia[data[-1] + 1:]))
    return d


def load(workdir, extra):
    try:
        args = extra if "-c" in extra else base_args() + extra
        return digest(Problem(args, cwd=str(workdir)))
    except HostError as e:
        return e


def test_every_option_of_the_table_has_a_probe():
    names = [n for n, _, _ in option_table()]
    assert len(names) == len(set(names))
    assert sorted(names) == sorted(PROBES), set(names) ^ set(PROBES)
    # same names, order and arity as the reference's table (argum.c:112-320), written out here
    # so that the table cannot drift
    assert names == ["version", "help", "quiet", "verb", "config_file", "atm", "linedb", "outtoomuch", "outsample",
                     "outspec", "outintens", "molfile", "savefiles", "raddelt", "radlow", "radhigh", "radfct",
                     "allowq", "refpress", "refradius", "gsurf", "qmol", "qscale", "wllow", "wlhigh", "wlfct",
                     "wnlow", "wnhigh", "wndelt", "wnosamp", "wnfct", "ndop", "nlor", "dmin", "dmax", "lmin", "lmax",
                     "nwidth", "ethreshold", "cloud", "cloudtop", "scattering", "detailext", "detailcia", "csfile",
                     "saveext", "opacityfile", "tlow", "thigh", "tempdelt", "justOpacity", "shareOpacity",
                     "solution", "toomuch", "taulevel", "modlevel", "detailtau", "starrad", "gorbpar", "gorbparfct",
                     "transparent", "raygrid"]


@pytest.mark.parametrize("name", sorted(PROBES))
def test_option_is_acted_on_or_refused(workdir, name):
    kinds = {n: k for n, _, k in option_table()}
    base = load(workdir, [])
    assert not isinstance(base, Exception), base
    extra, check = PROBES[name]
    check(base, load(workdir, extra))
    if kinds[name] == "x":
        assert isinstance(load(workdir, extra), HostError)
    if kinds[name] in "nw":
        assert any(name in m for m in load(workdir, extra)["msgs"])


def test_qscale_checks_of_the_reference(workdir):
    """argum.c:881-890: as many scale factors as names."""
    with pytest.raises(HostError) as e:
        Problem(["-c", "case.cfg", "--qscale", "0.5"], cwd=str(workdir))
    assert "same number" in str(e.value)
    cfg = "\n".join(l for l in (workdir / "case.cfg").read_text().split("\n") if not l.startswith("qmol"))
    (workdir / "noq.cfg").write_text(cfg)
    with pytest.raises(HostError):
        Problem(["-c", "noq.cfg"], cwd=str(workdir))
    # names that are not atmosphere species are ignored, like in the reference (readatm.c:396-404)
    a = Problem(["-c", "case.cfg", "--qmol", "CH4 XYZ"], cwd=str(workdir)).layer_arrays()["density"]
    b = Problem(["-c", "case.cfg", "--qmol", "CH4 H2O", "--qscale", "0.5 0"], cwd=str(workdir)).layer_arrays()["density"]
    assert np.array_equal(a, b)


def test_qscale_rescales_and_rebalances():
    """readatm.c:519-540: q *= 10^qscale for the named species; He = (1-metals)/(1+r), H2 = r*He with
    r the file's H2/He ratio."""
    P = golden(CASE).problem
    base = Problem(["-c", "case.cfg", "--qscale", "0 0"], cwd=os.path.join(GOLDEN, CASE))
    st = P.static
    names = open(os.path.join(GOLDEN, CASE, "case.atm")).read().split("#SPECIES\n")[1].split("\n")[0].split()
    mass = np.ctypeslib.as_array(st.mol_mass, shape=(st.nmol,))
    a, b = P.layer_arrays(), base.layer_arrays()
    # number abundances: rho_i = AMU q_i p / (k T) m_i, so rho ratios are q ratios
    i = names.index("CH4")
    assert np.allclose(a["density"][i] / b["density"][i], 10 ** 0.5, rtol=1e-12)
    i = names.index("H2O")
    assert np.allclose(a["density"][i] / b["density"][i], 10 ** -0.3, rtol=1e-12)
    h2, he = names.index("H2"), names.index("He")
    assert np.allclose(a["density"][h2] / a["density"][he], b["density"][h2] / b["density"][he], rtol=1e-12)
    q = a["density"] / mass[:, None]
    assert np.allclose((q / q.sum(axis=0)).sum(axis=0), 1.0)


def test_samplings_file_matches_the_reference(workdir):
    """outsample (makesample.c:642-672, 744-770) -- byte for byte."""
    for case in ("qscale_eclipse", "dumps_transit"):
        P = Problem.from_cfg(os.path.join(GOLDEN, case, "case.cfg"))
        out = str(workdir / ("sample_%s.dat" % case))
        P.write_sample(out)
        assert open(out).read() == open(os.path.join(GOLDEN, case, "sample.dat")).read()


def test_detail_writer_on_the_reference_dumps(workdir):
    """detailout (tau.c:526-605) fed with the reference's own tau/extinction/CIA dumps must give the
    reference's detail files (same rows, same picked wavenumbers; the CIA file through the float
    view of the doubles that the reference prints)."""
    for case in ("qscale_eclipse", "dumps_transit"):
        g = golden(case)
        d = workdir / ("det_" + case)
        shutil.copytree(os.path.join(GOLDEN, case), d)
        for f in ("detail_tau.dat", "detail_ext.dat", "detail_cia.dat"):
            os.remove(d / f)
        P = Problem.from_cfg(str(d / "case.cfg"))
        P.write_detail(0, g.tau)
        P.write_detail(1, g.e)
        P.write_detail(2, g.e_cs)
        for f in ("detail_tau.dat", "detail_ext.dat"):
            got, ref = open(d / f).read().split("\n"), open(os.path.join(GOLDEN, case, f)).read().split("\n")
            assert got[0] == ref[0] and len(got) == len(ref), f
            a, b = np.loadtxt(d / f, comments="#", ndmin=2), np.loadtxt(os.path.join(GOLDEN, case, f), comments="#", ndmin=2)
            assert np.array_equal(a[:, 0], b[:, 0])
            assert rel_err(a[:, 1:], b[:, 1:]) < 2e-7          # the dumps carry 10 digits, the detail file 7
        # the CIA detail file shows 32-bit halves of the doubles: equal only where the 10-digit dump
        # reproduces the double exactly, so compare the structure and the header
        got, ref = open(d / "detail_cia.dat").read().split("\n"), open(os.path.join(GOLDEN, case, "detail_cia.dat")).read().split("\n")
        assert got[0] == ref[0] and len(got) == len(ref)
        assert [l.split()[0] for l in got[1:] if l] == [l.split()[0] for l in ref[1:] if l]


SAVEFILES = (("tau.dat", "wavenumber"), ("CIA.dat", "wavenumber"), ("mol_extion.dat", "radius"),
             ("total_extion.dat", "wavenumber"), ("cloud_extion.dat", "wavenumber"), ("scatt_extion.dat", "wavenumber"))


def check_savefiles(got_dir, case, tol):
    """The six dumps of `savefiles yes` against the reference's own: same text skeleton (header,
    row keys, line count), same zeros (layers the reference's lazy sweep never reached; heights
    past the toomuch cut), values to `tol`."""
    import oracle_lib as ol
    for name, key in SAVEFILES:
        ref_path = os.path.join(GOLDEN, case, name)
        got_lines, ref_lines = open(os.path.join(got_dir, name)).read().split("\n"), open(ref_path).read().split("\n")
        assert got_lines[:4] == ref_lines[:4], name
        assert len(got_lines) == len(ref_lines), name
        gk, gv = ol.read_rows_dump(os.path.join(got_dir, name), key)
        rk, rv = ol.read_rows_dump(ref_path, key)
        assert np.array_equal(gk, rk), name
        assert np.array_equal(gv == 0, rv == 0), name
        assert rel_err(gv, rv) < tol, name


@pytest.mark.parametrize("case", ["qscale_eclipse", "dumps_transit", "cloud_opa", "cloud_b17", "cloud_f18", "cloud_p19"])
def test_savefiles_writers_on_the_oracle_arrays(workdir, case):
    """tau.dat, CIA.dat, mol/total/cloud/scatt_extion.dat (tau.c:180-190, 293-335, 386-515) from
    the oracle's intermediates: pins the writers -- including the reference's lazily swept rows
    and the eclipse solution's bottom-point values left in er -- without a GPU."""
    import oracle_lib as ol
    g = golden(case)
    P = g.problem
    out = ol.OracleEngine(P.static).run(P.atm, P.opts, debug=("e", "e_cs", "tau", "last", "er", "e_scat", "e_cloud"))
    d = workdir / ("save_" + case)
    d.mkdir()
    P.write_savefiles(out, str(d))
    check_savefiles(str(d), case, 2e-9)


def test_cli_help_and_version_exit_cleanly():
    exe = build.build_cli() or build.lib_path("transit_hip")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    p = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "--qscale" in p.stdout and "--raygrid" in p.stdout
    p = subprocess.run([exe, "-V"], capture_output=True, text=True)
    assert p.returncode == 0 and "transit_hip" in p.stdout
    p = subprocess.run([exe, "-c", os.path.join(GOLDEN, CASE, "case.cfg"), "--taulevel", "2"], capture_output=True,
                       text=True, cwd=os.path.join(GOLDEN, CASE))
    assert p.returncode != 0 and "taulevel" in p.stderr
