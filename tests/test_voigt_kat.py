"""The Voigt table pinned DIRECTLY against the reference's voigtn() (pu/src/voigt.c:369-483):
tests/golden/voigt/voigt_kat.json holds its float32 output for 33 (alphaD, alphaL, spacing)
combinations formed as getprofile() forms them (extinction.c:8-57) -- regime A (quick, > 99999
points), B (fine grid: mean of the bin edges), C (coarse grid: Simpson mean), the clamp to the
band, the three-point minimum, the aliasing rule of calcprofiles (opacity.c:262-265).
  * the CPU restatement must reproduce every float bit for bit;
  * the HIP table kernels (double instead of long double in the Region-I series) within 1 ulp.
"""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from cases import GOLDEN
from transit_amd import synth
from transit_amd.host import Problem

KAT = json.load(open(os.path.join(GOLDEN, "voigt", "voigt_kat.json")))


def expected(rec):
    """(indices, float32 values) the fixture holds for one profile."""
    if "values_u32" in rec:
        v = np.array(rec["values_u32"], dtype=np.uint32).view(np.float32)
        return np.arange(v.size), v
    return np.array(rec["sample_idx"]), np.array(rec["sample_u32"], dtype=np.uint32).view(np.float32)


def test_fixture_covers_the_regimes():
    recs = [(g, p) for g in KAT.values() for p in g["profiles"] if "nv" in p]
    assert len(recs) >= 30
    assert any(p["quick"] for _, p in recs)                                               # regime A
    dwn = lambda g: float.fromhex(g["dwn"])
    assert any(not p["quick"] and dwn(g) < float.fromhex(p["alphaD"]) / 49 for g, p in recs)    # regime B
    assert any(dwn(g) >= float.fromhex(p["alphaD"]) / 49 for g, p in recs)                # regime C
    assert any(p["nv"] == 3 for _, p in recs)                                             # at least three points
    assert any(p["nv"] == 2 * g["nown"] + 1 for g, p in recs)                             # clamp to the band
    assert any("alias_of" in p for g in KAT.values() for p in g["profiles"])


@pytest.mark.parametrize("grid", sorted(KAT))
def test_oracle_reproduces_voigtn_bit_for_bit(grid):
    lib = ol.oracle_library()
    for rec in KAT[grid]["profiles"]:
        if "nv" not in rec:
            continue
        nv = rec["nv"]
        out = np.zeros(nv, dtype=np.float32)
        rc = lib.trxo_voigt_profile(nv, float.fromhex(rec["half"]), float.fromhex(rec["alphaL"]), float.fromhex(rec["alphaD"]),
                                    out.ctypes.data_as(C.POINTER(C.c_float)), 1 if rec["quick"] else 0)
        assert rc == 1
        assert zlib.crc32(out.tobytes()) == rec["crc32"], (grid, rec["i"], rec["j"])
        idx, v = expected(rec)
        assert np.array_equal(out[idx].view(np.uint32), v.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("grid", sorted(KAT))
def test_hip_table_within_one_ulp_of_voigtn(tmp_path, grid):
    from transit_amd.engine import Engine
    g = KAT[grid]
    cfg = g["cfg"]
    d = str(tmp_path / grid)
    synth.make_case(d, nlines=200, wnlow=cfg["wnlow"], wnhigh=cfg["wnhigh"], wndelt=cfg["wndelt"], wnosamp=cfg["wnosamp"],
                    nlayers=8, solution="eclipse", seed=3, nwidth=float(cfg["nwidth"]),
                    extra={"ndop": "2", "nlor": "2", "dmin": repr(cfg["dmin"]), "dmax": repr(cfg["dmax"]),
                           "lmin": repr(cfg["lmin"]), "lmax": repr(cfg["lmax"])})
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn == g["nwn"] and P.static.nown == g["nown"]
    eng = Engine(P.static)
    ps, off, tab = eng.table()
    eng.close()
    worst = 0.0
    for rec in g["profiles"]:
        i, j = rec["i"], rec["j"]
        if "alias_of" in rec:                                     # opacity.c:262-265: the same profile object
            a, b = rec["alias_of"]
            assert off[i, j] == off[a, b] and ps[i, j] == ps[a, b]
            continue
        assert ps[i, j] == rec["nv"] // 2                         # getprofile's return value
        got = tab[off[i, j]: off[i, j] + rec["nv"]]
        idx, v = expected(rec)
        ulp = np.abs(got[idx].astype(np.float64) - v.astype(np.float64)) / np.spacing(np.abs(v)).astype(np.float64)
        worst = max(worst, float(ulp.max()))
        assert ulp.max() <= 1.0, (grid, i, j, float(ulp.max()))
        assert (ulp > 0).mean() < 0.02                            # and all but a few entries are identical
    assert worst <= 1.0
