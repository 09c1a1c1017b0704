"""Host side (C++: transit_amd/csrc/host): the reference's CLI/cfg surface, file
readers and samplings -- everything transit_init() does before the spectrum path."""
import os
import shutil

import numpy as np
import pytest

from cases import GOLDEN, golden
from transit_amd import synth
from transit_amd.host import HostError, Problem

CASE = os.path.join(GOLDEN, "eclipse_small")


def copy_case(tmp_path, name="eclipse_small"):
    d = tmp_path / name
    shutil.copytree(os.path.join(GOLDEN, name), d)
    return d


def test_defaults_match_the_reference_table():
    """argum.c:112-320 default strings."""
    P = Problem(["--atm", "case.atm", "--linedb", "case.tli", "--molfile", "molecules.dat",
                 "--wnlow", "2500", "--wnhigh", "2510", "--wndelt", "1", "--wnfct", "1"], cwd=CASE)
    assert P.option("toomuch") == "20" and P.opts.toomuch == 20.0
    assert P.opts.ethresh == 1e-8
    assert P.static.osamp == 2160 and P.static.ndop == 60 and P.static.nlor == 60
    assert np.float32(P.static.dmin) == np.float32(1e-3) and np.float32(P.static.lmax) == np.float32(10.0)
    assert P.static.timesalpha == 20.0
    assert P.opts.solution == 0 and P.opts.nangles == 5
    assert P.static.ncia == 0                      # no csfile given
    assert P.nwn == 11 and P.static.nown == 10 * 2160 + 1


def test_cfg_tokens_are_matched_as_prefixes_first_hit_wins(tmp_path):
    """procopt.c:675: strncmp(name, token, len(token)) in table order."""
    d = copy_case(tmp_path)
    cfg = (d / "case.cfg").read_text()
    assert "ethresh 1e-50" in cfg                  # the demo cfg itself relies on 'ethresh' -> 'ethreshold'
    import re
    cfg = cfg.replace("solution eclipse", "sol eclipse")
    cfg = re.sub(r"toomuch \S+", "toom 7.5", cfg)
    (d / "case.cfg").write_text(cfg)
    P = Problem.from_cfg(str(d / "case.cfg"))
    assert P.opts.ethresh == 1e-50 and P.opts.toomuch == 7.5 and P.opts.solution == 0
    # 'wn' is a prefix of wnlow, wnhigh, wndelt, wnosamp, wnfct: the first in the table wins
    (d / "case.cfg").write_text(cfg + "wn 2501\n")
    assert Problem.from_cfg(str(d / "case.cfg")).static.wn_i == 2501.0


def test_command_line_overrides_the_cfg_in_order(tmp_path):
    d = copy_case(tmp_path)
    P = Problem(["-c", "case.cfg", "--toomuch", "3", "--solution", "transit"], cwd=str(d))
    assert P.opts.toomuch == 3.0 and P.opts.solution == 1 and P.opts.nangles == 0
    P = Problem(["--toomuch", "3", "-c", "case.cfg"], cwd=str(d))     # cfg comes later: it wins
    assert P.opts.toomuch == 10.0


def test_wavelength_limits_give_the_reference_grid(tmp_path):
    """makesample.c:318-363: wn_i = 1/(wlhigh*wlfct), n = ((1+1e-8) f - i)/d + 1."""
    d = copy_case(tmp_path)
    cfg = (d / "case.cfg").read_text()
    cfg = "\n".join(l for l in cfg.split("\n") if not l.startswith(("wnlow", "wnhigh", "wnfct")))
    (d / "case.cfg").write_text(cfg + "\nwlhigh 4.0\nwllow 3.90625\n")
    P = Problem.from_cfg(str(d / "case.cfg"))
    assert P.static.wn_i == 1.0 / (4.0 * 1e-4)
    f = 1.0 / (3.90625 * 1e-4)
    assert P.nwn == int(((1.0 + 1e-8) * f - P.static.wn_i) / 1.0 + 1)
    assert P.static.nown == (P.nwn - 1) * 2160 + 1


def test_errors_are_codes_not_exits(tmp_path):
    d = copy_case(tmp_path)
    with pytest.raises(HostError) as e:
        Problem(["-c", "nosuch.cfg"], cwd=str(d))
    assert e.value.code == -1
    with pytest.raises(HostError):
        Problem(["-c", "case.cfg", "--frobnicate", "1"], cwd=str(d))
    with pytest.raises(HostError):
        Problem(["-c", "case.cfg", "--solution", "sideways"], cwd=str(d))
    with pytest.raises(HostError):
        Problem(["-c", "case.cfg", "--ethreshold", "0"], cwd=str(d))
    with pytest.raises(HostError):                       # argum.c:812: nwidth < 1
        Problem(["-c", "case.cfg", "--nwidth", "0.5"], cwd=str(d))
    with pytest.raises(HostError) as e:                  # crosssec.c:251-259: CIA must cover the band
        Problem(["-c", "case.cfg", "--wnhigh", "9000"], cwd=str(d))
    assert e.value.code in (-1, -5)
    (d / "molecules.dat").write_text("# empty\n 105 H2 2.01588 2.89 02 0.787 x\n")
    with pytest.raises(HostError):                       # readatm.c:697-703: unknown species
        Problem.from_cfg(str(d / "case.cfg"))


def test_top_down_atmosphere_is_resorted(tmp_path):
    """readatm.c:583-617."""
    d = copy_case(tmp_path)
    lines = (d / "case.atm").read_text().split("\n")
    k = next(i for i, l in enumerate(lines) if l.startswith("#Radius")) + 1
    data = [l for l in lines[k:] if l.strip()]
    (d / "case.atm").write_text("\n".join(lines[:k] + data[::-1]) + "\n")
    a = Problem.from_cfg(str(d / "case.cfg")).layer_arrays()
    b = golden("eclipse_small").problem.layer_arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key


def test_mass_abundances_use_the_mean_molecular_mass(tmp_path):
    """transit.h:58-69 (stateeqnford) and readatm.c:137-146 (checkaddmm)."""
    d = copy_case(tmp_path)
    atm = synth.demo_atmosphere(30)
    masses = {m[1]: m[2] for m in synth.MOLECULE_TABLE}
    mvec = np.array([masses[s] for s in atm.species])
    qn = atm.abundance
    mu = (qn * mvec).sum(axis=1, keepdims=True)
    atm_m = synth.Atmosphere(atm.species, atm.radius, atm.pressure, atm.temperature,
                             np.array([[float("%.4e" % v) for v in row] for row in qn * mvec / mu]), by_mass=True)
    synth.write_atm(str(d / "case.atm"), atm_m)
    a = Problem.from_cfg(str(d / "case.cfg")).layer_arrays()
    q = atm_m.abundance
    mm = 1.0 / (q / mvec).sum(axis=1)
    p, t = atm.pressure * 1e6, atm.temperature
    rho = 1.66053886e-24 * q.T * p / 1.380658e-16 / t * mm
    assert np.max(np.abs(a["density"][:, :-1] / rho[:, :-1] - 1)) < 1e-14     # top layer carries spline round-off


def test_tli_range_selection_drops_lines_outside_the_band(tmp_path):
    """readlineinfo.c:496-526: per-isotope bracket of [1/wn_f, 1/wn_i]."""
    d = copy_case(tmp_path)
    P = Problem(["-c", "case.cfg", "--wnlow", "2520", "--wnhigh", "2540"], cwd=str(d))
    full = golden("eclipse_small").problem.static.nlines
    n = P.static.nlines
    assert 0 < n < full
    wl = np.ctypeslib.as_array(P.static.wl_um, shape=(n,))
    wn = 1e4 / wl
    iso = np.ctypeslib.as_array(P.static.isoid, shape=(n,))
    for b in np.unique(iso):                         # at most one neighbour beyond each end per isotope
        w = wn[iso == b]
        assert np.sum(w > 2540) <= 1 and np.sum(w < 2520) <= 1
        assert np.all(np.diff(w) <= 0)               # wavelength ascending = wavenumber descending


def _tli_counts_offset(raw):
    """Byte offset of (nlines int64, nisol int32, counts int64[nisol]) in a TLI v6 file."""
    import struct
    pos = 4 + 6 + 16
    (ndb,) = struct.unpack_from("<H", raw, pos); pos += 2
    for _ in range(ndb):
        for _ in range(2):
            (n,) = struct.unpack_from("<H", raw, pos); pos += 2 + n
        nT, nI = struct.unpack_from("<2H", raw, pos); pos += 4 + 8 * nT
        for _ in range(nI):
            (n,) = struct.unpack_from("<H", raw, pos); pos += 2 + n + 16 + 8 * nT
    return pos


@pytest.mark.parametrize("damage", ["negative_total", "negative_count", "sum_mismatch", "huge_total", "too_many_isotopes",
                                    "truncated", "no_temperatures"])
def test_malformed_tli_headers_are_refused_not_followed(tmp_path, damage):
    """The line blocks are mapped and indexed through the header's counts: a header whose counts
    do not add up must end in TRX_E_ARG, never in a read outside the mapping."""
    import struct
    d = copy_case(tmp_path)
    tli = [f for f in os.listdir(d) if f.endswith(".tli")][0]
    raw = bytearray(open(d / tli, "rb").read())
    o = _tli_counts_offset(raw)
    nlines, nisol = struct.unpack_from("<qi", raw, o)
    cnt = list(struct.unpack_from("<%dq" % nisol, raw, o + 12))
    assert sum(cnt) == nlines and len(raw) == o + 12 + 8 * nisol + 26 * nlines
    if damage == "negative_total":
        struct.pack_into("<q", raw, o, -5)
    elif damage == "negative_count":
        struct.pack_into("<q", raw, o + 12, -cnt[0])
    elif damage == "sum_mismatch":
        struct.pack_into("<q", raw, o + 12, cnt[0] + 3)
    elif damage == "huge_total":
        struct.pack_into("<q", raw, o, 2**61)
    elif damage == "too_many_isotopes":
        struct.pack_into("<i", raw, o + 8, 10**6)
    elif damage == "truncated":
        raw = raw[: len(raw) - 26 * nlines // 3]
    elif damage == "no_temperatures":
        pos = 4 + 6 + 16 + 2
        for _ in range(2):
            (n,) = struct.unpack_from("<H", raw, pos); pos += 2 + n
        struct.pack_into("<H", raw, pos, 0)
    open(d / tli, "wb").write(bytes(raw))
    with pytest.raises(HostError) as e:
        Problem.from_cfg(str(d / "case.cfg"))
    assert e.value.code == -1                            # TRX_E_ARG


def test_multi_database_tli_and_species_mapping():
    P = golden("multi_species").problem
    st = P.static
    assert st.niso == 6
    imol = np.ctypeslib.as_array(st.iso_imol, shape=(6,))
    species = synth.DEMO_SPECIES
    assert [species[i] for i in imol] == ["H2O", "H2O", "H2O", "CH4", "CH4", "CO"]
    z = P.layer_arrays()["zpart"]
    assert z.shape == (6, P.nlayer) and np.all(z > 0)


def test_shard_bounds_balance_the_work_not_the_bins(tmp_path):
    """trh_shard_bounds / shard.balanced_bounds: with a line list three times denser in
    the upper half of the band the cuts move so that every rank gets the same work within 10 %."""
    import ctypes as C
    from transit_amd import host as H
    from transit_amd.shard import balanced_bounds, bin_costs, all_bounds
    d = str(tmp_path / "dense")
    lo = synth.synth_linedb(20000, 2500, 2750, seed=3)
    hi = synth.synth_linedb(60000, 2750, 3000, seed=4, name="HITEMP CO (synthetic)", molname="CO",
                            iso_names=("26",), iso_masses=(27.994915,), iso_ratios=(0.98654,), iso_split=(1.0,), z_scale=107.0)
    synth.make_case(d, wnlow=2500, wnhigh=3000, nlayers=30, solution="eclipse", dbs=[lo, hi])
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    cost = bin_costs(P.static, P.nlayer)
    for world in (2, 3, 8):
        b = balanced_bounds(cost, world)
        assert b[0][0] == 0 and b[-1][1] == P.nwn and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        work = np.array([cost[l:h].sum() for l, h in b])
        assert work.max() / work.mean() < 1.1
        even = np.array([cost[l:h].sum() for l, h in all_bounds(P.nwn, world)])
        assert even.max() / even.mean() > work.max() / work.mean()        # the equal-bins split is worse
        # the C++ host side cuts at the same places
        lib = H._load()
        lib.trh_shard_bounds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        lib.trh_shard_bounds.restype = C.c_int
        out = (C.c_int64 * (world + 1))()
        assert lib.trh_shard_bounds(P._h, world, out) == 0
        assert list(out) == [b[0][0]] + [h for _, h in b]
    with pytest.raises(ValueError):
        balanced_bounds(cost, P.nwn + 1)
