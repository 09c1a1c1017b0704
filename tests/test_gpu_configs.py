"""BASELINE.json configs[2..4] at their FULL sizes on the GPU, checked through
size-independent properties plus oracle parity on what the CPU finishes in seconds:

  C3  H2O+CH4+CO, 1-30 um at 1 cm-1 (9667 points), 200 layers, 3 x 10^6 lines
  C4  transmission geometry, H2-H2 + H2-He CIA, wavenumber axis in 8 shards
  C5  ~10^7 points (dnu = 9.667e-4, wnosamp 1) x 150 layers x 10^7 lines

(configs[0..1] are tests/test_gpu_properties.py and the goldens.)"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from cases import rel_err
from transit_amd import synth
from transit_amd.engine import Engine
from transit_amd.host import Problem
from transit_amd.shard import all_bounds
from tolerances import DEBUG_KEYS, assert_tau_close

pytestmark = pytest.mark.gpu


three_species = synth.three_species_dbs


# Optical depths: the arithmetic's 1e-9 plus the rounding steps of the reference's parabola in
# absolute radius where the extinction jumps between layers -- tests/tolerances.py, demonstrated
# on the CPU in tests/test_tolerance_mechanism.py.


def hip_and_oracle(cfg):
    Q = Problem.from_cfg(cfg)
    hip = Engine(Q.static)
    got = hip.run(Q.atm, Q.opts, debug=True)
    hip.close()
    ora = ol.OracleEngine(Q.static)
    ref = ora.run(Q.atm, Q.opts, debug=DEBUG_KEYS)
    ora.close()
    return Q, got, ref


def check_cut(P, tau, last):
    crossed = tau[np.arange(P.nwn), last] > P.opts.toomuch
    assert np.all(crossed | (last == P.nlayer - 1))
    for w in range(0, P.nwn, max(P.nwn // 400, 1)):
        t = tau[w, : last[w] + 1]
        assert np.all(np.isfinite(t)) and np.all(np.diff(t) >= -1e-12 * max(t[-1], 1e-300))
        assert np.all(tau[w, last[w] + 1:] == 0)


def stitched(P, world, pick=None):
    """Spectrum of the `world`-way wavenumber split, one shard (one engine) at a time --
    what the N ranks of a multi-GPU job compute, minus the final gather."""
    parts = {}
    try:
        for r, (lo, hi) in enumerate(all_bounds(P.nwn, world)):
            if pick is not None and r not in pick:
                continue
            P.set_shard(lo, hi)
            e = Engine(P.static)
            parts[r] = (lo, hi, e.run(P.atm, P.opts)["spectrum"])
            e.close()
    finally:
        P.set_shard(0, P.nwn)
    return parts


# ---- C3 -------------------------------------------------------------------------------
def test_c3_three_species_full_size(tmp_path):
    kw = dict(wnlow=333.33, wnhigh=10000, wndelt=1.0, wnosamp=2160, nlayers=200, solution="eclipse",
              toomuch=10.0, ethresh=1e-50)
    d = str(tmp_path / "c3")
    synth.make_case(d, dbs=three_species(1_000_000, 333.33, 10000), **kw)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn == 9667 and P.nlayer == 200 and P.static.nlines == 3_000_000 and P.static.niso == 6
    eng = Engine(P.static)
    out = eng.run(P.atm, P.opts, debug=("tau", "last"))
    st = eng.stats()
    eng.close()
    assert 2_999_000 < st["nlines_inrange"] <= 3_000_000        # the grid ends 0.67 cm-1 below wnhigh
    assert np.all(np.isfinite(out["spectrum"])) and np.all(out["spectrum"] > 0)
    check_cut(P, out["tau"], out["last"])
    parts = stitched(P, 3)
    whole = np.concatenate([parts[r][2] for r in range(3)])
    assert rel_err(whole, out["spectrum"]) < 1e-10

    # the same grid, layers and species with 2 % of the lines against the oracle
    d2 = str(tmp_path / "c3_sub")
    synth.make_case(d2, dbs=three_species(20_000, 333.33, 10000), **kw)
    Q, got, ref = hip_and_oracle(os.path.join(d2, "case.cfg"))
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert_tau_close(Q, got, ref)


# ---- C4 -------------------------------------------------------------------------------
def test_c4_transmission_two_cia_eight_shards(tmp_path):
    kw = dict(wnlow=2500, wnhigh=5000, wndelt=1.0, wnosamp=2160, nlayers=100, solution="transit",
              toomuch=10.0, ethresh=1e-50, ncia=2, seed=1234)
    d = str(tmp_path / "c4")
    synth.make_case(d, nlines=1_000_000, **kw)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.static.ncia == 2 and P.nwn == 2501
    eng = Engine(P.static)
    out = eng.run(P.atm, P.opts, debug=("tau", "last"))
    eng.close()
    mod = out["spectrum"]
    assert np.all(np.isfinite(mod)) and np.all(mod > 0) and np.all(mod < 1)      # (Rp/Rs)^2-like
    check_cut(P, out["tau"], out["last"])
    parts = stitched(P, 8)
    whole = np.concatenate([parts[r][2] for r in range(8)])
    # R^2 - 2*integral cancellation amplifies the re-association of the per-bin sums
    assert rel_err(whole, mod) < 1e-9

    d2 = str(tmp_path / "c4_sub")
    synth.make_case(d2, nlines=20_000, **kw)
    Q, got, ref = hip_and_oracle(os.path.join(d2, "case.cfg"))
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-8
    assert_tau_close(Q, got, ref)


# ---- C5 -------------------------------------------------------------------------------
C5 = dict(wndelt=9.667e-4, wnosamp=1, nlayers=150, solution="eclipse", toomuch=10.0, ethresh=1e-50)


def test_c5_resolution_against_oracle(tmp_path):
    """C5's grid spacing, layer count and line density per cm-1 on an 8 cm-1 window."""
    d = str(tmp_path / "c5_win")
    synth.make_case(d, nlines=8_000, wnlow=3000, wnhigh=3008, seed=77, **C5)
    Q, got, ref = hip_and_oracle(os.path.join(d, "case.cfg"))
    assert Q.nwn > 8_000
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert_tau_close(Q, got, ref)
    sw = got["computed"].astype(bool)
    assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-9


def test_c5_full_size_runs_and_shards_stitch(tmp_path):
    d = str(tmp_path / "c5")
    synth.make_case(d, nlines=10_000_000, wnlow=333.33, wnhigh=10000, seed=1234, **C5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn > 9_990_000 and P.nlayer == 150 and P.static.nlines == 10_000_000
    eng = Engine(P.static)
    out = eng.run(P.atm, P.opts, debug=("last",))
    st = eng.stats()
    eng.close()
    spec, last = out["spectrum"], out["last"]
    assert 9_999_000 < st["nlines_inrange"] <= 10_000_000 and st["layers_swept"] > 100
    assert np.all(np.isfinite(spec)) and np.all(spec > 0)
    assert last.min() >= 0 and last.max() < 150
    # three of the eight shards an 8-GPU job would run (first, an inner one, last)
    for r, (lo, hi, part) in stitched(P, 8, pick=(0, 3, 7)).items():
        assert rel_err(part, spec[lo:hi]) < 1e-10, r
