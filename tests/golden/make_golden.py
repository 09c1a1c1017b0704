#!/usr/bin/env python3
"""Regenerate the golden fixtures: synthetic inputs -> compiled reference -> outputs.

Run in the build container only (needs oracle/_ref/transit, built by
`make -C oracle ref` from the reference sources where they lie).  Each case
directory holds the inputs (cfg, atmosphere, TLI, CIA, molecules) and what the
reference wrote for them: spectrum.dat, toomuch.dat and the --savefiles dumps
tau.dat / mol_extion.dat / CIA.dat (tau.c:180-190, 293-329).  Only data is
kept; no reference source or binary enters the repository.

    python tests/golden/make_golden.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from transit_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "transit")
REF_REENTRY = os.path.join(ROOT, "oracle", "_ref", "transit_reentry")   # oracle/ref_reentry_main.c
# the same reference with tau.c (unmodified) compiled by clang -ftrivial-auto-var-init=zero (oracle/Makefile):
# defined output for the cloud models that read tau.c's uninitialised mean_dens[] (cloud flags 2-5)
REF_ZINIT = os.path.join(ROOT, "oracle", "_ref", "transit_zinit")

KEEP = ["case.cfg", "case.atm", "case.tli", "molecules.dat", "cia_h2h2.dat", "cia_h2he.dat",
        "spectrum.dat", "toomuch.dat", "tau.dat", "mol_extion.dat", "CIA.dat", "intens.dat",
        # cases that ask for them: samplings file, detailout files, the three per-wavenumber dumps
        "sample.dat", "detail_ext.dat", "detail_tau.dat", "detail_cia.dat",
        "total_extion.dat", "cloud_extion.dat", "scatt_extion.dat", "ext.sav"]
DUMPS3 = ("total_extion.dat", "cloud_extion.dat", "scatt_extion.dat")

CASES = {
    # demo-shaped emission run, narrow band
    "eclipse_small": dict(nlines=3000, wnlow=2500, wnhigh=2560, nlayers=30, solution="eclipse",
                          extra={"outintens": "intens.dat"}),      # + the per-angle intensity file (printintens)
    # --saveext: the reference's extinction save file (savefile_extinct, extinction.c:62-95) of a small
    # transmission run -- its lazy sweep reaches only the layers its rays need
    "saveext_transit": dict(nlines=1500, wnlow=2500, wnhigh=2530, nlayers=20, solution="transit", seed=808,
                            extra={"saveext": "ext.sav"}),
    # transmission geometry, two CIA tables
    "transit_small": dict(nlines=3000, wnlow=2500, wnhigh=2560, nlayers=30, solution="transit", ncia=2,
                          seed=4321),
    # coarse fine-grid + high threshold: co-adding, skipped lines, sticky Doppler index
    "coadd_thresh": dict(nlines=20000, wnlow=2500, wnhigh=2540, nlayers=24, wnosamp=24, ethresh=1e-6,
                         solution="eclipse", seed=99, toomuch=8.0),
    # grey cloud deck + Rayleigh-like scattering, lines reaching beyond the band edges
    "cloud_scatter": dict(nlines=2000, wnlow=3000, wnhigh=3040, nlayers=40, solution="transit",
                          seed=7, line_margin=3.0,
                          extra={"cloudtop": "-2.0", "scattering": "1.5"}),
    # The parametrised cloud types opa/B17/F18/P19 (cloud flags 2-5, extinction.c:630-693): the reference accumulates
    # their mean density into an uninitialised array (tau.c:129,203) and its gcc build returns NaN for all four;
    # `transit_zinit` is the same sources with tau.c compiled by clang -ftrivial-auto-var-init=zero (for `--cloud ext`
    # the two builds agree to the last printed digit in both geometries)
    "cloud_opa": dict(nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution="eclipse", seed=7, zinit=True, keep_dumps3=True,
                      extra={"cloud": "opa,1e-3,-3.0,0.5"}),
    "cloud_b17": dict(nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution="transit", seed=7, zinit=True, keep_dumps3=True,
                      extra={"cloud": "B17,1e-7,-3.0,0.5,-1.5"}),
    "cloud_f18": dict(nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution="eclipse", seed=7, zinit=True, keep_dumps3=True,
                      extra={"cloud": "F18,1e-7,-3.0,0.5,2.0,0.8,1e-5"}),
    "cloud_p19": dict(nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution="transit", seed=7, zinit=True, keep_dumps3=True,
                      extra={"cloud": "P19,1e-9,-3.0,0.5,-2.0,1e-22,3000.0"}),
    # polarizability scattering (scattering flag 2, extinction.c:617-621) in emission geometry.
    "scat_polar": dict(nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution="eclipse", seed=8,
                       extra={"scattering": "polar"}),
    # opaque-disc modulation level and a transparent planet, odd layer count
    "transit_modm1": dict(nlines=4000, wnlow=4000, wnhigh=4030, nlayers=31, solution="transit", seed=11,
                          toomuch=5.0, extra={"modlevel": "-1"}),
    # high-resolution regime (BASELINE configs[4] in small): output grid far finer than
    # the lines, no oversampling -> every profile spans hundreds of bins; the Voigt
    # table is built with the coarse-grid Simpson averaging (voigt.c:427-433)
    "highres_fine": dict(nlines=1500, wnlow=2500, wnhigh=2502, wndelt=0.002, wnosamp=1, nlayers=24,
                         solution="eclipse", seed=5, toomuch=10.0),
    # oversampling 4 on a 0.02 cm-1 grid: wide profiles with a phase-major table
    # three databases, six isotopes, threshold active (the per-layer maximum is taken
    # over all species at once, extinction.c:421-426 with permol = 0)
    "multi_species": dict(wnlow=1000, wnhigh=1100, nlayers=22, solution="eclipse", ethresh=1e-7,
                          toomuch=12.0, dbs="multi"),
    # opacity-grid mode: the first run builds the grid file (calcopacity), writes it and
    # computes the spectrum from it (interpolmolext); kept as opacity_ref.dat
    "opacity_grid": dict(nlines=2500, wnlow=2500, wnhigh=2540, nlayers=16, solution="eclipse", seed=31,
                         ethresh=1e-6, dbs="multi2",
                         extra={"opacityfile": "opac.dat", "tlow": "1100", "thigh": "1900", "tempdelt": "200"}),
    # library API re-entry (BART): transit_init once, run_transit(T, abundances) three times
    # (reloadatm + radpress + makeradsample, readatm.c:722-865); see oracle/ref_reentry_main.c
    "reentry": dict(nlines=2000, wnlow=2500, wnhigh=2540, nlayers=20, solution="eclipse", seed=55,
                    reentry=True, extra={"refpress": "0.1", "gsurf": "1000.0"}),
    # same in transmission geometry: the case that exposed the reference's compiled form of
    # tau.c:274, b = (h*hfct)*(1/rfct); the radii after every radpress()
    # are kept too (<prefix>N_radii.dat)
    # the same with the library's setters between the runs (set_radius, set_cloudtop, set_scattering:
    # transit.c:97-116) -- reentry_inputs.txt carries them as keyword lines
    "reentry_set": dict(nlines=2000, wnlow=2500, wnhigh=2540, nlayers=20, solution="eclipse", seed=55,
                        reentry="set", extra={"refpress": "0.1", "gsurf": "1000.0"}),
    "reentry_set_transit": dict(nlines=2000, wnlow=2500, wnhigh=2540, nlayers=20, solution="transit", seed=55,
                                reentry="set", extra={"refpress": "0.1", "gsurf": "1000.0"}),
    "reentry_transit": dict(nlines=2000, wnlow=2500, wnhigh=2540, nlayers=20, solution="transit", seed=55,
                            reentry=True, extra={"refpress": "0.1", "gsurf": "1000.0"}),
    # abundance scaling while the atmosphere is read (qmol/qscale, readatm.c:394-405, 519-540:
    # the named species are multiplied by 10^qscale, H2 and He re-balanced), emission geometry;
    # also every optional output file: samplings, detailout at a few wavenumbers, total/cloud/
    # scattering extinction dumps (eclipse geometry leaves its edited bottom values in them)
    "qscale_eclipse": dict(nlines=2500, wnlow=2500, wnhigh=2540, nlayers=26, solution="eclipse", seed=61,
                           keep_dumps3=True,
                           extra={"qmol": "CH4 H2O", "qscale": "0.5 -0.3", "outsample": "sample.dat",
                                  "scattering": "1.2",
                                  "detailext": "detail_ext.dat:2503.2,2520,2539.9",
                                  "detailtau": "detail_tau.dat:2500,2531.5",
                                  "detailcia": "detail_cia.dat:2510,2540"}),
    # the same outputs in transmission geometry with a cloud deck and scattering
    "dumps_transit": dict(nlines=2000, wnlow=3000, wnhigh=3030, nlayers=25, solution="transit", seed=62,
                          keep_dumps3=True, ncia=2,
                          extra={"cloudtop": "-1.5", "scattering": "1.5", "outsample": "sample.dat",
                                 "detailext": "detail_ext.dat:3001,3015.5", "detailtau": "detail_tau.dat:3029.9",
                                 "detailcia": "detail_cia.dat:3000,3030"}),
    # radius resampling (makesample.c:144-300, 409-549): layers every 400 km instead of the
    # atmosphere file's own sampling, T / p / abundances / partition functions splined onto them
    "resample_radius": dict(nlines=2000, wnlow=2500, wnhigh=2530, nlayers=25, solution="eclipse", seed=71,
                            extra={"raddelt": "400"}),
    "resample_transit": dict(nlines=2000, wnlow=2500, wnhigh=2530, nlayers=25, solution="transit", seed=72, ncia=2,
                             extra={"raddelt": "350", "radlow": "93500"}),
    # 130 isotopes in 8 line databases (the reference allocates per isotope without a cap,
    # readlineinfo.c:134-224; its own isotopologues.dat lists more than 100)
    "many_isotopes": dict(dbs="iso130", wnlow=2500, wnhigh=2520, nlayers=12, solution="eclipse", seed=91),
    "midres_os4": dict(nlines=2500, wnlow=3100, wnhigh=3108, wndelt=0.02, wnosamp=4, nlayers=20,
                       solution="transit", seed=17, ncia=2),
}


def multi_species_dbs():
    """Three line databases (H2O, CH4, CO) with 3 + 2 + 1 isotopes: BASELINE configs[2] in small."""
    h2o = synth.synth_linedb(1800, 1000, 1100, seed=21, name="HITEMP H2O (synthetic)", molname="H2O",
                             iso_names=("161", "181", "171"), iso_masses=(18.010565, 20.014811, 19.01478),
                             iso_ratios=(0.997317, 0.002, 0.000372), iso_split=(0.8, 0.15, 0.05), z_scale=170.0)
    ch4 = synth.synth_linedb(1500, 1000, 1100, seed=22)
    co = synth.synth_linedb(700, 1000, 1100, seed=23, name="HITEMP CO (synthetic)", molname="CO",
                            iso_names=("26",), iso_masses=(27.994915,), iso_ratios=(0.98654,), iso_split=(1.0,),
                            z_scale=107.0, log_gf=(-10.0, -4.0))
    return [h2o, ch4, co]


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/transit is missing: run `make -C oracle ref` first")
    only = set(sys.argv[1:])
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        d = os.path.join(HERE, name)
        tmp = d + ".tmp"
        shutil.rmtree(tmp, ignore_errors=True)
        kw = dict(kw)
        if kw.get("dbs") == "multi":
            kw["dbs"] = multi_species_dbs()
        if kw.get("dbs") == "iso130":
            kw["dbs"] = synth.many_isotope_dbs()
        if kw.get("dbs") == "multi2":      # two molecules, three isotopes, in the 2500-2540 band
            kw["dbs"] = [synth.synth_linedb(1500, 2500, 2540, seed=41),
                         synth.synth_linedb(900, 2500, 2540, seed=42, name="HITEMP CO (synthetic)", molname="CO",
                                            iso_names=("26",), iso_masses=(27.994915,), iso_ratios=(0.98654,),
                                            iso_split=(1.0,), z_scale=107.0, log_gf=(-10.0, -4.0))]
        extra = dict(kw.pop("extra", {}))
        extra.update({"savefiles": "yes"})
        reentry = kw.pop("reentry", False)
        keep3 = kw.pop("keep_dumps3", False)
        exe = REF_ZINIT if kw.pop("zinit", False) else REF
        if not os.path.exists(exe):
            sys.exit("%s is missing: run `make -C oracle ref` first" % exe)
        if reentry:
            import numpy as np
            atm = synth.demo_atmosphere(kw["nlayers"])
            k0 = int(np.argmin(np.abs(atm.pressure - 0.1)))
            extra["refradius"] = "%.3f" % atm.radius[k0]
            kw["atm"] = atm
        synth.make_case(tmp, extra=extra, **kw)
        if reentry:
            nl, nm = len(atm.radius), len(atm.species)
            runs = []
            q = atm.abundance.T.copy()                     # [nmol][nlayer]
            runs.append(np.concatenate([atm.temperature, q.ravel()]))
            q2 = q.copy(); q2[atm.species.index("CH4")] *= 3.0
            runs.append(np.concatenate([atm.temperature + 50.0, q2.ravel()]))
            q3 = q.copy(); q3[atm.species.index("H2O")] *= 0.5
            t3 = atm.temperature * (1.0 - 0.04 * np.linspace(0, 1, nl))
            runs.append(np.concatenate([t3, q3.ravel()]))
            setters = {}
            if reentry == "set":          # before run 2 / 3 / 4 (run 4 repeats the first atmosphere under all three)
                runs.append(runs[0])
                setters = {1: "radius %.17g" % (1.015 * atm.radius[k0]), 2: "cloudtop -1.5", 3: "scattering 1 1.25"}
            with open(os.path.join(tmp, "reentry_inputs.txt"), "w") as f:
                for k, r in enumerate(runs):
                    if k in setters:
                        f.write(setters[k] + "\n")
                    f.write(" ".join("%.17g" % v for v in r) + "\n")
        if reentry:
            log = subprocess.run([REF_REENTRY, "case.cfg", "reentry_inputs.txt", "reentry_out"], cwd=tmp,
                                 capture_output=True, text=True)
            if log.returncode != 0:
                sys.exit("reference re-entry driver failed:\n%s\n%s" % (log.stdout[-2000:], log.stderr[-2000:]))
        log = subprocess.run([exe, "-c", "case.cfg"], cwd=tmp, capture_output=True, text=True)
        if log.returncode != 0:
            sys.exit("reference failed on %s:\n%s\n%s" % (name, log.stdout[-2000:], log.stderr[-2000:]))
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        for f in KEEP:
            if f in DUMPS3 and not keep3:
                continue
            if os.path.exists(os.path.join(tmp, f)):
                shutil.copy(os.path.join(tmp, f), os.path.join(d, f))
        for f in sorted(os.listdir(tmp)):
            if f.startswith("reentry_"):
                shutil.copy(os.path.join(tmp, f), os.path.join(d, f))
        if os.path.exists(os.path.join(tmp, "opac.dat")):
            shutil.copy(os.path.join(tmp, "opac.dat"), os.path.join(d, "opacity_ref.dat"))
        shutil.rmtree(tmp)
        print("%-16s -> %s" % (name, ", ".join(sorted(os.listdir(d)))))


if __name__ == "__main__":
    main()
