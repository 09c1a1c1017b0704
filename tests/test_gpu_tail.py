"""k_ray_tail (trx_tail.hip.h: a hinted run's combines, optical depths and emission / modulation as ONE
kernel behind its walks) against the step kernels it stands for (TRX_RAY_TAIL=0): the same
operations on the same values, so the same BITS in every output -- extinction, edited extinction,
optical depth, stopping heights, intensities, spectrum -- on goldens, on a demo-shaped case (two
walk steps), on shards, with CIA, and when the atmosphere moves under the remembered depth (rays
go deeper: the tail's pass is resumed by the step kernels; shallower: a shorter plan)."""
import os

import numpy as np
import pytest

from cases import GOLDEN
from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu

KEYS = ("e", "e_cs", "tau", "last", "intens", "computed", "er", "e_scat", "e_cloud")


def engines(P):
    os.environ["TRX_RAY_TAIL"] = "0"
    try:
        ref = Engine(P.static)
    finally:
        os.environ.pop("TRX_RAY_TAIL", None)
    return Engine(P.static), ref


def assert_same(ra, rb, note=None):
    sw = ra["computed"].astype(bool) if "computed" in ra else slice(None)
    assert set(ra) == set(rb)
    for k in ra:
        if k in ("e", "e_cs", "e_scat", "e_cloud"):
            assert np.array_equal(ra[k][sw], rb[k][sw]), (k, note)
        elif k == "er":
            # (edited extinction: written down to the ray's stopping height only)
            nl = ra[k].shape[0]
            used = (nl - 1 - np.arange(nl))[:, None] <= ra["last"][None, :]
            assert np.array_equal(ra[k][used], rb[k][used]), (k, note)
        elif k == "tau":
            # heights a ray never reached hold whatever an earlier run left: compare what the ray used
            last = ra["last"]
            used = np.arange(ra[k].shape[1])[None, :] <= last[:, None]
            assert np.array_equal(ra[k][used], rb[k][used]), (k, note)
        else:
            assert np.array_equal(ra[k], rb[k]), (k, note)


def tail_runs(msgs):
    return sum("ray tail over" in m for m in msgs)


@pytest.mark.parametrize("case", ["eclipse_small", "coadd_thresh", "multi_species", "many_isotopes", "midres_os4", "scat_polar", "qscale_eclipse",
                                  "transit_small", "cloud_scatter", "resample_transit", "transit_modm1", "dumps_transit"])
def test_tail_on_goldens(case):
    P = Problem.from_cfg(os.path.join(GOLDEN, case, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    t, r = engines(P)
    try:
        for k in range(3):                               # unhinted, hinted with dumps, hinted production
            dbg = KEYS if k < 2 else None
            a = t.run(P.atm, P.opts, debug=dbg) if dbg else t.run(P.atm, P.opts)
            n_tail = tail_runs(msgs)
            b = r.run(P.atm, P.opts, debug=dbg) if dbg else r.run(P.atm, P.opts)
            assert tail_runs(msgs) == n_tail             # (the reference engine never takes the tail)
            assert_same(a, b, (case, k))
        assert t.stats()["layers_swept"] == r.stats()["layers_swept"]
    finally:
        engine.set_log(None)
        t.close(); r.close()


@pytest.mark.parametrize("ncia,nlayers,solution", [(0, 100, "eclipse"), (2, 100, "eclipse"), (1, 40, "eclipse"), (0, 7, "eclipse"),
                                                   (2, 100, "transit"), (0, 40, "transit"), (1, 9, "transit")])
def test_tail_demo_shape_moving_atmosphere_and_shards(tmp_path, ncia, nlayers, solution):
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2800, wndelt=1.0, wnosamp=2160, nlayers=nlayers,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=11, ncia=ncia)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    dens = np.ctypeslib.as_array(P.atm.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    t, r = engines(P)
    try:
        # same atmosphere (hint exact), thinner (rays go deeper: resumed), the same again, denser
        # (rays stop higher: a shorter plan next time), back
        for k, sc in enumerate([1.0, 1.0, 1.0, 0.03, 0.03, 0.03, 40.0, 40.0, 1.0, 1.0]):
            dens[:] = base * sc
            dbg = KEYS if k % 3 != 2 else None
            a = t.run(P.atm, P.opts, debug=dbg) if dbg else t.run(P.atm, P.opts)
            b = r.run(P.atm, P.opts, debug=dbg) if dbg else r.run(P.atm, P.opts)
            assert_same(a, b, (k, sc))
            assert t.stats()["layers_swept"] == r.stats()["layers_swept"], (k, sc)
        assert tail_runs(msgs) >= 6, tail_runs(msgs)
        dens[:] = base
        full = t.run(P.atm, P.opts)["spectrum"]
    finally:
        dens[:] = base
        t.close(); r.close()
    # shards: ragged last block (nsh % 8 != 0), a one-ray shard
    for lo, hi in [(37, 211), (0, 13), (P.nwn - 1, P.nwn)]:
        P.set_shard(lo, hi)
        try:
            t, r = engines(P)
            try:
                for k in range(3):
                    a, b = t.run(P.atm, P.opts, debug=KEYS), r.run(P.atm, P.opts, debug=KEYS)
                    assert_same(a, b, (lo, hi, k))
                assert np.array_equal(a["spectrum"], full[lo:hi]), (lo, hi)
            finally:
                t.close(); r.close()
        finally:
            P.set_shard(0, P.nwn)
    engine.set_log(None)
    assert tail_runs(msgs) >= 12


def test_tail_with_copy_commands_instead_of_pinned_outputs(tmp_path):
    """TRX_TAIL_DIRECT=0: the tail writes spectrum and flags to device memory and copy commands bring
    them back (the form a run graph or a device-side spectrum uses) -- the same bits, the same depth hint."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=60_000, wnlow=2500, wnhigh=2700, wndelt=1.0, wnosamp=2160, nlayers=90,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=3, ncia=1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    os.environ["TRX_TAIL_DIRECT"] = "0"
    try:
        c = Engine(P.static)
    finally:
        os.environ.pop("TRX_TAIL_DIRECT", None)
    t, r = engines(P)
    dens = np.ctypeslib.as_array(P.atm.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()
    try:
        for k, sc in enumerate([1.0, 1.0, 0.05, 0.05, 1.0, 1.0]):
            dens[:] = base * sc
            a, b, x = t.run(P.atm, P.opts), c.run(P.atm, P.opts), r.run(P.atm, P.opts)
            assert np.array_equal(a["spectrum"], x["spectrum"]) and np.array_equal(b["spectrum"], x["spectrum"]), (k, sc)
            assert t.stats()["layers_swept"] == r.stats()["layers_swept"] == c.stats()["layers_swept"], (k, sc)
    finally:
        dens[:] = base
        t.close(); r.close(); c.close()


@pytest.mark.parametrize("solution,ncia", [("eclipse", 1), ("transit", 2)])
def test_two_queues_against_one(tmp_path, solution, ncia):
    """A hinted run of two walk steps puts its second walk on the side queue next to the first and the tail
    behind it there (TRX_TWO_QUEUES=0: one queue, the walks one behind the other): the same bits, the same
    depth hint, also when the atmosphere moves under the remembered depth (rays go deeper: the pass is resumed
    by the step kernels with the flags the host has added up) -- and with the kernels bracketed by events
    (trx_opts.profile 1: the same plan and the same bits; one queue: the walks' span is at least the sum of their durations)."""
    d = str(tmp_path / "q")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2800, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=11, ncia=ncia)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    os.environ["TRX_TWO_QUEUES"] = "0"
    try:
        one = Engine(P.static)
    finally:
        os.environ.pop("TRX_TWO_QUEUES", None)
    two, ref = engines(P)
    dens = np.ctypeslib.as_array(P.atm.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    spans = 0
    try:
        for k, sc in enumerate([1.0, 1.0, 1.0, 0.05, 0.05, 1.0, 1.0, 3.0, 3.0]):
            dens[:] = base * sc
            P.opts.profile = 1 if k in (2, 6) else 0
            a, b, x = two.run(P.atm, P.opts), one.run(P.atm, P.opts), ref.run(P.atm, P.opts)
            assert np.array_equal(a["spectrum"], x["spectrum"]) and np.array_equal(b["spectrum"], x["spectrum"]), (k, sc)
            sa, sb, sx = two.stats(), one.stats(), ref.stats()
            # (the tail has swept both steps of its plan when it looks at the rays; the step kernels of a plan whose first
            # step closes every ray -- the atmosphere got denser -- gate the second step off and do not count it)
            assert sa["layers_swept"] == sb["layers_swept"] >= sx["layers_swept"], (k, sc)
            if P.opts.profile and sa["walk_steps"] == 2:
                assert sa["ms_walk_span"] > 0 and sa["ms_k_walk"] > 0, (k, sa["ms_walk_span"], sa["ms_k_walk"])      # (short walks: the second may start after the first has ended)
                assert sb["ms_walk_span"] >= 0.999 * sb["ms_k_walk"] > 0, (k, sb["ms_walk_span"], sb["ms_k_walk"])     # one queue: one behind the other
                spans += 1
    finally:
        P.opts.profile = 0
        dens[:] = base
        engine.set_log(None)
        two.close(); one.close(); ref.close()
    assert tail_runs(msgs) >= 8
    assert spans >= 1


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_tail_without_the_per_bin_record_table(tmp_path, solution):
    """TRX_NO_BINREC: the tail finds a bin's records through the ranges' numbers (blo, off) instead of the plan's
    per-bin record table -- what a handle does whose shard is too large for that table (niso x bins > 2^17) or
    whose records pass 2^31.  The same bits as the default form and as the step kernels."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=60_000, wnlow=2500, wnhigh=2700, wndelt=1.0, wnosamp=2160, nlayers=90,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=3, ncia=2 if solution == "transit" else 1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    os.environ["TRX_NO_BINREC"] = "1"
    try:
        c = Engine(P.static)
    finally:
        os.environ.pop("TRX_NO_BINREC", None)
    t, r = engines(P)
    try:
        for k in range(3):
            a, b, x = t.run(P.atm, P.opts, debug=KEYS), c.run(P.atm, P.opts, debug=KEYS), r.run(P.atm, P.opts, debug=KEYS)
            assert_same(a, x, k); assert_same(b, x, k)
    finally:
        t.close(); r.close(); c.close()
        engine.set_log(None)
    assert tail_runs(msgs) >= 4          # (hinted runs of both tail handles)


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
@pytest.mark.parametrize("extra", [{"cloudtop": "-2.0", "scattering": "1.5"}, {"scattering": "polar"}])
def test_tail_with_cloud_and_scattering_models(tmp_path, extra, solution):
    """extinction.c:587-693 switched on: the tail adds the layer parts of both models to the line and
    CIA extinction in the order of tau.c:231-232, like the step kernels."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=60_000, wnlow=2500, wnhigh=2700, wndelt=1.0, wnosamp=2160, nlayers=90,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=3, ncia=1, extra=extra)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    t, r = engines(P)
    try:
        for k in range(3):
            a, b = t.run(P.atm, P.opts, debug=KEYS), r.run(P.atm, P.opts, debug=KEYS)
            assert_same(a, b, (extra, k))
    finally:
        engine.set_log(None)
        t.close(); r.close()
    assert tail_runs(msgs) >= 2


def test_tail_with_twelve_angles(tmp_path):
    """More than eight emission angles: the instantiation that keeps state for all 16 (k_ray_tail<16>)."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=40_000, wnlow=2500, wnhigh=2650, wndelt=1.0, wnosamp=2160, nlayers=70,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=9, ncia=1,
                    raygrid="0 7 14 21 28 35 42 49 56 63 70 77")
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.opts.nangles == 12
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    t, r = engines(P)
    try:
        for k in range(3):
            a, b = t.run(P.atm, P.opts, debug=KEYS), r.run(P.atm, P.opts, debug=KEYS)
            assert_same(a, b, k)
    finally:
        engine.set_log(None)
        t.close(); r.close()
    assert tail_runs(msgs) >= 2
