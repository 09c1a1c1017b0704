"""Synthetic inputs in the reference's own on-disk formats.

Everything the spectrum path reads (TLI line list, atmosphere, CIA table,
molecule table, cfg file) can be generated here so that tests, bench.py and the
reference binary (oracle/_ref/transit, test infrastructure only) consume
byte-identical files.  Formats follow the readers of the reference:

* TLI v6 binary ........ transit/src/readlineinfo.c:88-244, 416-537
* atmosphere text ...... transit/src/readatm.c:277-428, 444-620
* CIA / cross-section .. transit/src/crosssec.c:87-233
* molecule table ....... transit/src/readatm.c:626-717
* cfg .................. pu/src/procopt.c:651-705 ("name value" lines)

Recipe for the synthetic line list: SURVEY.md section 8(d) (seed 1234, PCG64).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

TLI_MAGIC = bytes([0xFF, 0xFF - ord("I"), 0xFF - ord("L"), 0xFF - ord("T")])

# cgs constants used only to lay out a hydrostatic synthetic atmosphere
_KB = 1.380658e-16
_AMU = 1.66053886e-24

# (ID, name, mass g/mol, diameter Angstrom, source tag, polarizability A^3, long name)
MOLECULE_TABLE = [
    (101, "H2O", 18.01528, 3.2, "01", 1.501, "Water"),
    (102, "CH4", 16.0425, 4.0, "01", 2.448, "Methane"),
    (103, "CO", 28.0101, 2.8, "01", 1.953, "Carbon Monoxide"),
    (104, "CO2", 44.0095, 2.8, "01", 2.507, "Carbon Dioxide"),
    (105, "H2", 2.01588, 2.89, "02", 0.787, "Molecular Hydrogen"),
    (106, "NH3", 17.03052, 3.6, "01", 2.103, "Ammonia"),
    (110, "N2", 28.01340, 3.64, "02", 1.710, "Molecular Nitrogen"),
    (1, "H", 1.007940, 2.4, "01", 0.667, "Hydrogen"),
    (2, "He", 4.0026020, 2.0, "01", 0.208, "Helium"),
    (6, "C", 12.0107, 1.7, "04", 1.760, "Carbon"),
    (7, "N", 14.0067, 1.55, "04", 1.100, "Nitrogen"),
    (8, "O", 15.9994, 1.52, "04", 0.802, "Oxygen"),
]

DEMO_SPECIES = ["H", "He", "C", "N", "O", "H2", "CO", "CO2", "CH4", "H2O"]
DEMO_ABUND = [1e-10, 0.15, 1e-10, 1e-10, 1e-10, 0.84969, 1e-4, 1e-4, 1e-5, 1e-4]


# --------------------------------------------------------------------------
# molecule table
# --------------------------------------------------------------------------
def write_molfile(path: str) -> str:
    with open(path, "w") as f:
        f.write("# Molecular info (synthetic table in the transit molecules.dat grammar)\n")
        f.write("# ID    Molecule  Mass         Diameter  Diameter  Polarizability Long\n")
        f.write("#       Name      g/mol        Angstrom  source    Angstrom^3     name\n")
        for mid, name, mass, diam, src, pol, longname in MOLECULE_TABLE:
            f.write(" %3d    %-8s  %-12.7g %-9.4g %-9s %-14.4g %s\n"
                    % (mid, name, mass, diam, src, pol, longname))
    return path


# --------------------------------------------------------------------------
# atmosphere
# --------------------------------------------------------------------------
@dataclass
class Atmosphere:
    species: List[str]
    radius: np.ndarray        # [nlayer] in units of ur (km when ur = 1e5)
    pressure: np.ndarray      # [nlayer] in units of up (bar when up = 1e6)
    temperature: np.ndarray   # [nlayer] K
    abundance: np.ndarray     # [nlayer, nspecies]
    ur: float = 1e5
    up: float = 1e6
    by_mass: bool = False


def demo_atmosphere(nlayers: int = 100, p_bottom: float = 100.0, p_top: float = 1e-5,
                    t_bottom: float = 1651.63, t_top: float = 1237.47,
                    r_bottom_km: float = 92173.413, gravity: float = 1000.0,
                    species: Sequence[str] = DEMO_SPECIES,
                    abundances: Sequence[float] = DEMO_ABUND) -> Atmosphere:
    """Demo-shaped atmosphere by formula: log-uniform pressures (bottom first),
    a smooth hot-to-cool temperature step and hydrostatic radii."""
    logp = np.linspace(np.log10(p_bottom), np.log10(p_top), nlayers)
    p = 10.0 ** logp
    t = t_top + (t_bottom - t_top) * 0.5 * (1.0 + np.tanh((logp + 0.45) / 0.55))
    masses = {m[1]: m[2] for m in MOLECULE_TABLE}
    mu = sum(a * masses[s] for s, a in zip(species, abundances))
    r = np.empty(nlayers)
    r[0] = r_bottom_km
    for i in range(1, nlayers):
        tm = 0.5 * (t[i] + t[i - 1])
        scale_h = _KB * tm / (mu * _AMU * gravity) / 1e5   # km
        r[i] = r[i - 1] + scale_h * np.log(p[i - 1] / p[i])
    q = np.tile(np.asarray(abundances, dtype=float), (nlayers, 1))
    # round to what the text file will carry so that in-memory == on-disk
    r = np.round(r, 3)
    p = np.array([float("%.4e" % v) for v in p])
    t = np.round(t, 2)
    return Atmosphere(list(species), r, p, t, q)


def write_atm(path: str, atm: Atmosphere) -> str:
    with open(path, "w") as f:
        f.write("# Synthetic atmosphere in the transit atmosphere-file grammar\n")
        f.write("# Units: radius (ur cm), pressure (up barye), temperature (K)\n\n")
        f.write("ur %g\nup %g\nq %s\n\n" % (atm.ur, atm.up, "mass" if atm.by_mass else "number"))
        f.write("#SPECIES\n%s\n\n" % " ".join(atm.species))
        f.write("#TEADATA\n#Radius    Pressure   Temp  abundances\n")
        for i in range(len(atm.radius)):
            row = " %.3f %.4e %.2f " % (atm.radius[i], atm.pressure[i], atm.temperature[i])
            row += " ".join("%.4e" % v for v in atm.abundance[i])
            f.write(row + " \n")
    return path


# --------------------------------------------------------------------------
# line lists (TLI v6)
# --------------------------------------------------------------------------
@dataclass
class Isotope:
    name: str
    mass: float
    ratio: float
    z: np.ndarray             # partition function on the DB temperature grid


@dataclass
class LineDB:
    name: str
    molname: str
    temps: np.ndarray
    isotopes: List[Isotope]
    # per isotope line arrays, wavelength (um) ascending
    wl: List[np.ndarray] = field(default_factory=list)
    elow: List[np.ndarray] = field(default_factory=list)
    gf: List[np.ndarray] = field(default_factory=list)


def partition_function(temps: np.ndarray, scale: float = 590.0) -> np.ndarray:
    return scale * (temps / 296.0) ** 1.5 * (1.0 + (temps / 1500.0) ** 3)


def synth_linedb(nlines: int, wn_lo: float, wn_hi: float, seed: int = 1234,
                 name: str = "HITRAN CH4 (synthetic)", molname: str = "CH4",
                 iso_names: Sequence[str] = ("211", "311"),
                 iso_masses: Sequence[float] = (16.0313, 17.03466),
                 iso_ratios: Sequence[float] = (0.98827, 0.0111),
                 iso_split: Sequence[float] = (0.9, 0.1),
                 z_scale: float = 590.0,
                 log_gf: Sequence[float] = (-12.0, -5.0),
                 elow_max: float = 6000.0) -> LineDB:
    rng = np.random.default_rng(seed)
    temps = np.arange(70.0, 3000.0 + 1e-9, 10.0)
    isos = [Isotope(n, m, r, partition_function(temps, z_scale * (1.0 + 0.05 * k)))
            for k, (n, m, r) in enumerate(zip(iso_names, iso_masses, iso_ratios))]
    db = LineDB(name, molname, temps, isos)
    counts = [int(round(nlines * s)) for s in iso_split]
    counts[0] += nlines - sum(counts)
    for c in counts:
        wn = np.sort(rng.uniform(wn_lo, wn_hi, c))[::-1]      # descending wn
        db.wl.append(1e4 / wn)                                 # ascending wavelength (um)
        db.elow.append(rng.uniform(0.0, elow_max, c))
        db.gf.append(10.0 ** rng.uniform(log_gf[0], log_gf[1], c))
    return db


def many_isotope_dbs(niso_total: int = 130, lines_per_iso: int = 40, wn_lo: float = 2500.0, wn_hi: float = 2520.0,
                     seed: int = 900) -> List[LineDB]:
    """Eight line databases (one per molecule of the demo atmosphere that can carry lines) with
    `niso_total` isotopes between them: more than any fixed small table holds."""
    mols = ["H2O", "CH4", "CO", "CO2", "H2", "H", "He", "C"]
    masses = {m[1]: m[2] for m in MOLECULE_TABLE}
    per = [niso_total // len(mols) + (1 if k < niso_total % len(mols) else 0) for k in range(len(mols))]
    dbs = []
    for k, (mol, n) in enumerate(zip(mols, per)):
        ratios = np.array([0.5 ** (i + 1) for i in range(n)])
        ratios[0] += 1.0 - ratios.sum()
        dbs.append(synth_linedb(n * lines_per_iso, wn_lo, wn_hi, seed=seed + k, name="synthetic %s, %d isotopes" % (mol, n),
                                molname=mol, iso_names=tuple("%d%02d" % (k + 1, i) for i in range(n)),
                                iso_masses=tuple(masses[mol] + 0.37 * i for i in range(n)), iso_ratios=tuple(float(r) for r in ratios),
                                iso_split=tuple([1.0 / n] * n), z_scale=100.0 + 60.0 * k, log_gf=(-9.0, -4.0)))
    return dbs


def write_tli(path: str, dbs: Sequence[LineDB], wl_ini: Optional[float] = None,
              wl_fin: Optional[float] = None) -> str:
    """TLI v6, little-endian, no padding; lines sorted by (cumulative) isotope
    then ascending wavelength, four SoA blocks (pylineread.py:403-418)."""
    all_wl = np.concatenate([w for db in dbs for w in db.wl]) if dbs else np.zeros(0)
    if wl_ini is None:
        wl_ini = float(np.floor(all_wl.min() * 100) / 100) if all_wl.size else 1.0
    if wl_fin is None:
        wl_fin = float(np.ceil(all_wl.max() * 100) / 100) if all_wl.size else 2.0
    with open(path, "wb") as f:
        f.write(TLI_MAGIC)
        f.write(struct.pack("<3H", 6, 6, 0))
        f.write(struct.pack("<2d", wl_ini, wl_fin))
        f.write(struct.pack("<H", len(dbs)))
        for db in dbs:
            for s in (db.name, db.molname):
                b = s.encode()
                f.write(struct.pack("<H", len(b)) + b)
            f.write(struct.pack("<2H", len(db.temps), len(db.isotopes)))
            f.write(np.asarray(db.temps, "<f8").tobytes())
            for iso in db.isotopes:
                b = iso.name.encode()
                f.write(struct.pack("<H", len(b)) + b)
                f.write(struct.pack("<2d", iso.mass, iso.ratio))
                f.write(np.asarray(iso.z, "<f8").tobytes())
        counts, wl, isoid, elow, gf = [], [], [], [], []
        cum = 0
        for db in dbs:
            for k in range(len(db.isotopes)):
                n = len(db.wl[k])
                counts.append(n)
                wl.append(db.wl[k]); elow.append(db.elow[k]); gf.append(db.gf[k])
                isoid.append(np.full(n, cum + k, dtype="<i2"))
            cum += len(db.isotopes)
        ntot = int(sum(counts))
        f.write(struct.pack("<q", ntot))
        f.write(struct.pack("<i", len(counts)))
        f.write(np.asarray(counts, "<i8").tobytes())
        if ntot:
            f.write(np.concatenate(wl).astype("<f8").tobytes())
            f.write(np.concatenate(isoid).astype("<i2").tobytes())
            f.write(np.concatenate(elow).astype("<f8").tobytes())
            f.write(np.concatenate(gf).astype("<f8").tobytes())
    return path


# --------------------------------------------------------------------------
# CIA tables
# --------------------------------------------------------------------------
def synth_cia(wn_lo: float, wn_hi: float, dwn: float = 20.0,
              temps: Sequence[float] = (400, 500, 600, 700, 800, 900, 1000, 2000, 3000,
                                        4000, 5000, 6000, 7000),
              amp: float = 2e-6, centre: float = 4200.0, width: float = 600.0):
    """Smooth H2-H2-like collision-induced band (cm-1 amagat-2)."""
    wn = np.arange(wn_lo, wn_hi + 0.5 * dwn, dwn)
    t = np.asarray(temps, dtype=float)
    band = np.exp(-0.5 * ((wn[:, None] - centre) / (width * (1 + t[None, :] / 6000.0))) ** 2)
    cs = amp * (t[None, :] / 1000.0) ** 0.5 * band + 1e-10 * (1.0 + wn[:, None] / 1e4)
    return wn, t, cs


def write_cia(path: str, species: Sequence[str], wn: np.ndarray, temps: np.ndarray,
              cs: np.ndarray) -> str:
    with open(path, "w") as f:
        f.write("# Synthetic collision-induced absorption table (transit CS grammar)\n\n")
        f.write("i %s\n" % " ".join(species))
        f.write("t " + " ".join("%10g" % t for t in temps) + "\n\n")
        f.write("# Wavenumber in cm-1, CIA coefficients in cm-1 amagat-%d:\n" % len(species))
        for i in range(len(wn)):
            f.write("%10.2f " % wn[i] + " ".join("%.4e" % v for v in cs[i]) + "\n")
    return path


# --------------------------------------------------------------------------
# cfg
# --------------------------------------------------------------------------
def write_cfg(path: str, options: Dict[str, object]) -> str:
    with open(path, "w") as f:
        f.write("# transit configuration (name value)\n")
        for k, v in options.items():
            if v is None:
                continue
            f.write("%s %s\n" % (k, v))
    return path


# --------------------------------------------------------------------------
# one-call case builder
# --------------------------------------------------------------------------
def three_species_dbs(n_each: int, lo: float, hi: float) -> List[LineDB]:
    """H2O + CH4 + CO line databases of n_each lines (six isotopes): BASELINE configs[2]'s multi-species TLI
    (three databases over 333.33-10000 cm-1)."""
    h2o = synth_linedb(n_each, lo, hi, seed=21, name="HITEMP H2O (synthetic)", molname="H2O",
                       iso_names=("161", "181", "171"), iso_masses=(18.010565, 20.014811, 19.01478),
                       iso_ratios=(0.997317, 0.002, 0.000372), iso_split=(0.8, 0.15, 0.05), z_scale=170.0)
    ch4 = synth_linedb(n_each, lo, hi, seed=22)
    co = synth_linedb(n_each, lo, hi, seed=23, name="HITEMP CO (synthetic)", molname="CO",
                      iso_names=("26",), iso_masses=(27.994915,), iso_ratios=(0.98654,), iso_split=(1.0,),
                      z_scale=107.0, log_gf=(-10.0, -4.0))
    return [h2o, ch4, co]


def make_case(outdir: str, *, nlines: int = 3000, wnlow: float = 2500.0, wnhigh: float = 2560.0,
              wndelt: float = 1.0, wnosamp: int = 2160, nlayers: int = 30,
              solution: str = "eclipse", toomuch: float = 10.0, ethresh: float = 1e-50,
              nwidth: float = 20.0, raygrid: str = "0 20 40 60 80", ncia: int = 1,
              seed: int = 1234, extra: Optional[Dict[str, object]] = None,
              dbs: Optional[Sequence[LineDB]] = None,
              atm: Optional[Atmosphere] = None, line_margin: float = 0.0) -> Dict[str, str]:
    """Write atm/TLI/CIA/molfile/cfg for one run into *outdir* (relative file
    names inside the cfg, so the directory can be moved)."""
    os.makedirs(outdir, exist_ok=True)
    if atm is None:
        atm = demo_atmosphere(nlayers)
    if dbs is None:
        dbs = [synth_linedb(nlines, wnlow - line_margin, wnhigh + line_margin, seed)]
    paths = {
        "atm": write_atm(os.path.join(outdir, "case.atm"), atm),
        "linedb": write_tli(os.path.join(outdir, "case.tli"), dbs),
        "molfile": write_molfile(os.path.join(outdir, "molecules.dat")),
    }
    cia_files = []
    pad = 2 * 20.0
    wn, t, cs = synth_cia(max(20.0, wnlow - pad - (wnlow % 20.0)), wnhigh + pad)
    if ncia >= 1:
        cia_files.append(write_cia(os.path.join(outdir, "cia_h2h2.dat"), ["H2", "H2"], wn, t, cs))
    if ncia >= 2:
        cia_files.append(write_cia(os.path.join(outdir, "cia_h2he.dat"), ["H2", "He"], wn,
                                   t[4:], 0.3 * cs[:, 4:]))
    opts: Dict[str, object] = {
        "atm": "case.atm", "linedb": "case.tli", "molfile": "molecules.dat",
        "csfile": ",".join(os.path.basename(c) for c in cia_files) if cia_files else None,
        "wnlow": wnlow, "wnhigh": wnhigh, "wndelt": wndelt, "wnosamp": wnosamp, "wnfct": 1.0,
        "wlfct": 1e-4,
        "solution": solution, "raygrid": raygrid if solution == "eclipse" else None,
        "toomuch": toomuch, "ethresh": ethresh, "nwidth": nwidth,
        "verb": 2, "outspec": "spectrum.dat", "outtoomuch": "toomuch.dat",
    }
    if extra:
        opts.update(extra)
    paths["cfg"] = write_cfg(os.path.join(outdir, "case.cfg"), opts)
    paths["dir"] = outdir
    return paths
