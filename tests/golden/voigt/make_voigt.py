#!/usr/bin/env python3
"""Known-answer vectors for the Voigt table straight from the reference's voigtn()
(pu/src/voigt.c:369-483), called through oracle/_ref/libpu_ref.so in the build container.

For every grid below (a 2x2 width grid, a fine-grid spacing, a band) the arguments getprofile()
(transit/src/extinction.c:8-57) would hand to voigtn are formed -- float widths, nvgt with its
"< 2 -> 3" and "> 2*nwave -> 2*nwave+1" rules, VOIGT_QUICK above 99999 points -- and voigtn's
float32 output is recorded: in full for profiles up to 2048 points, else as a CRC32 of its bytes
plus a fixed sample of points.  Covers regime A (quick, > 99999 points), B (fine grid, mean of
the bin edges), C (coarse grid, Simpson mean), the clamp to the band, the 3-point minimum and
the aliasing rule of calcprofiles (opacity.c:262-265: entry (i,j) reuses (i-1,j) when
aDop[i]*10 < aLor[j]).

    python tests/golden/voigt/make_voigt.py      ->  tests/golden/voigt/voigt_kat.json
"""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

# name: band and grids (cfg options); every grid is 2 x 2
GRIDS = {
    "demo_narrow":   dict(wnlow=2500, wnhigh=2510, wndelt=1.0, wnosamp=2160, dmin=1e-3, dmax=0.25, lmin=1e-4, lmax=10.0, nwidth=20),
    "coarse_C":      dict(wnlow=2500, wnhigh=2520, wndelt=1.0, wnosamp=24, dmin=2e-3, dmax=0.05, lmin=1e-3, lmax=0.3, nwidth=20),
    "fine_B":        dict(wnlow=2500, wnhigh=2502, wndelt=0.002, wnosamp=2, dmin=0.06, dmax=0.25, lmin=1e-4, lmax=0.02, nwidth=20),
    "quick_A":       dict(wnlow=2500, wnhigh=2510, wndelt=1.0, wnosamp=5000, dmin=0.01, dmax=0.5, lmin=1e-3, lmax=0.2, nwidth=20),
    "band_clamp":    dict(wnlow=2500, wnhigh=2501, wndelt=0.01, wnosamp=2, dmin=1e-3, dmax=0.1, lmin=0.5, lmax=5.0, nwidth=20),
    "three_points":  dict(wnlow=2500, wnhigh=2600, wndelt=1.0, wnosamp=10, dmin=1e-4, dmax=2e-3, lmin=1e-4, lmax=1e-3, nwidth=20),
    "alias_rows":    dict(wnlow=2500, wnhigh=2510, wndelt=0.1, wnosamp=10, dmin=1e-3, dmax=0.01, lmin=0.05, lmax=1.0, nwidth=12),
    "wide_nwidth":   dict(wnlow=3000, wnhigh=3004, wndelt=0.05, wnosamp=4, dmin=5e-3, dmax=0.2, lmin=2e-3, lmax=0.5, nwidth=40),
    "lorentz_heavy": dict(wnlow=1000, wnhigh=1003, wndelt=0.02, wnosamp=3, dmin=2e-3, dmax=8e-3, lmin=0.02, lmax=0.07, nwidth=20),
}
SAMPLE_STRIDE = 251


def main():
    import oracle_lib as ol
    pu = ol.ref_pu_library()
    if pu is None:
        sys.exit("oracle/_ref/libpu_ref.so is missing: run `make -C oracle ref` first")
    out = {}
    for name, g in GRIDS.items():
        # makewnsample (makesample.c:80): n = (long)(((1+1e-8) f - i)/d + 1), fine grid (n-1)*o + 1
        nwn = int(((1.0 + 1e-8) * g["wnhigh"] - g["wnlow"]) / g["wndelt"] + 1)
        nown = (nwn - 1) * g["wnosamp"] + 1
        dwn = g["wndelt"] / g["wnosamp"]
        f32 = lambda v: float(np.float32(v))
        # logspace of the float hint fields (structures_tr.h:333; iomisc.c:1064-1083), two points
        adop = [f32(g["dmin"]), 10 ** (np.log10(f32(g["dmin"])) + (np.log10(f32(g["dmax"])) - np.log10(f32(g["dmin"]))))]
        alor = [f32(g["lmin"]), 10 ** (np.log10(f32(g["lmin"])) + (np.log10(f32(g["lmax"])) - np.log10(f32(g["lmin"]))))]
        ta = f32(g["nwidth"])
        profs = []
        for i in range(2):
            for j in range(2):
                if adop[i] * 10.0 < alor[j] and i != 0:                 # opacity.c:262-265
                    profs.append(dict(i=i, j=j, alias_of=[i - 1, j]))
                    continue
                dop, lor = f32(adop[i]), f32(alor[j])                    # getprofile's float parameters
                big = max(dop, lor)
                nvgt = 2 * int(big * ta / dwn + 0.5) + 1
                if nvgt < 2:
                    nvgt = 3
                if nvgt > 2 * nown:
                    nvgt = 2 * nown + 1
                half = dwn * (nvgt // 2)
                buf = (C.c_float * nvgt)()                               # the caller owns the profile (extinction.c:46)
                res = C.cast(buf, C.POINTER(C.c_float))
                rc = pu.voigtn(nvgt, half, lor, dop, C.byref(res), -1.0, 1 if nvgt > 99999 else 0)
                assert rc == 1, (name, i, j, rc)
                v = np.frombuffer(buf, dtype=np.float32).copy()
                rec = dict(i=i, j=j, nv=nvgt, half=float(half).hex(), alphaL=float(lor).hex(), alphaD=float(dop).hex(),
                           quick=bool(nvgt > 99999), crc32=zlib.crc32(v.tobytes()))
                if nvgt <= 2048:
                    rec["values_u32"] = v.view(np.uint32).tolist()
                else:
                    idx = sorted(set(list(range(0, 64)) + list(range(nvgt - 64, nvgt)) + list(range(nvgt // 2 - 64, nvgt // 2 + 65)) +
                                     list(range(0, nvgt, SAMPLE_STRIDE))))
                    rec["sample_idx"] = idx
                    rec["sample_u32"] = v.view(np.uint32)[idx].tolist()
                profs.append(rec)
        out[name] = dict(cfg=g, nwn=nwn, nown=nown, dwn=float(dwn).hex(), profiles=profs)
    with open(os.path.join(HERE, "voigt_kat.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    n = sum(1 for g in out.values() for p in g["profiles"] if "nv" in p)
    print("%d profiles in %d grids -> voigt_kat.json (%d bytes)" % (n, len(out), os.path.getsize(os.path.join(HERE, "voigt_kat.json"))))
    for name, g in out.items():
        print("  %-14s" % name, [(p.get("nv"), "quick" if p.get("quick") else "") if "nv" in p else "alias" for p in g["profiles"]])


if __name__ == "__main__":
    main()
