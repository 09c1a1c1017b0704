"""k_line_walk_lanes (steps of at most 32 layers with frames of 8+ bins: lanes = LINES for the strengths,
blocks of groups that share a cell for the accumulation; trx_lanes.hip.h) against k_line_walk /
k_line_walk_packed: the same lines in the same order, the same base points of the rebased exponential,
the same frame and the same partial records -- so the same BITS in extinction, optical depth and
spectrum.  Goldens (forced on their sparse lists: single-group blocks, frames that jump), demo-shaped
atmospheres whose deep step has 17 layers, a shard, lists with long co-added groups and several
isotopes."""
import os

import numpy as np
import pytest

from cases import GOLDEN
from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


def both(P, runs=2, debug=("e", "tau", "last", "computed"), env=None):
    out, used = [], 0
    for val in ("2", "0"):                 # lanes form wherever it applies (also on sparse lists) / never
        msgs = []
        engine.set_log(lambda lvl, m: msgs.append(m), 5)
        os.environ["TRX_LANES_WALK"] = val
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            e = Engine(P.static)
        finally:
            os.environ.pop("TRX_LANES_WALK", None)
            for k in (env or {}):
                os.environ.pop(k, None)
        res = [e.run(P.atm, P.opts, debug=debug) for _ in range(runs)]      # unhinted, then hinted (another step plan)
        res.append(e.run(P.atm, P.opts))                                      # and a production run (range skipping on)
        res.append(e.run(P.atm, P.opts))
        e.close()
        engine.set_log(None)
        if val == "2":
            used = sum("walk: lanes = lines" in m for m in msgs)
        else:
            assert not any("walk: lanes = lines" in m for m in msgs)
        out.append(res)
    return out[0], out[1], used


def assert_same(a, b):
    for ra, rb in zip(a, b):
        sw = ra["computed"].astype(bool) if "computed" in ra else slice(None)
        for k in ra:
            if k == "e":
                assert np.array_equal(ra[k][sw], rb[k][sw]), k
            else:
                assert np.array_equal(ra[k], rb[k]), k


@pytest.mark.parametrize("case", ["eclipse_small", "transit_small", "coadd_thresh", "cloud_scatter", "multi_species",
                                  "many_isotopes", "resample_transit", "midres_os4", "qscale_eclipse"])
def test_lanes_walk_on_goldens(case):
    P = Problem.from_cfg(os.path.join(GOLDEN, case, "case.cfg"))
    a, b, _ = both(P)
    assert_same(a, b)


@pytest.mark.parametrize("nlayers,solution", [(100, "eclipse"), (100, "transit"), (24, "transit"), (31, "eclipse")])
def test_lanes_walk_demo_shape(tmp_path, nlayers, solution):
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2800, wndelt=1.0, wnosamp=2160, nlayers=nlayers,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=11, ncia=2 if solution == "transit" else 1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a, b, used = both(P)
    assert used > 0, "the lanes form was never taken: the test compares nothing"
    assert_same(a, b)
    P.set_shard(37, 211)                       # a shard: only the ranges that reach it are launched
    try:
        sa, sb, used = both(P, runs=1)
    finally:
        P.set_shard(0, P.nwn)
    assert used > 0
    assert_same(sa, sb)
    assert np.array_equal(sa[-1]["spectrum"], a[-1]["spectrum"][37:211])


def test_lanes_walk_without_the_compact_rows(tmp_path):
    """TRX_NO_ROWS32: k_line_walk_lanes<8> gathers from the 64-byte rows of the walk's table copy (what a handle
    does whose compact copy would pass 4 GB or whose slab would pass 2^24 bytes) -- the same bits as with the
    32-byte rows and as k_line_walk."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=120_000, wnlow=2500, wnhigh=2800, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=11)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a, b, used = both(P, env={"TRX_NO_ROWS32": "1"})
    assert used > 0, "the lanes form was never taken: the test compares nothing"
    assert_same(a, b)
    c, _, used = both(P)                       # (the default form: compact rows)
    assert used > 0
    assert_same(a, c)


def test_lanes_walk_threshold_and_coadding(tmp_path):
    """A coarse fine grid (wnosamp 400: 2.5 lines per fine-grid point, co-added groups of up to ~10 lines)
    and a threshold that drops groups."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=60_000, wnlow=2500, wnhigh=2560, wndelt=1.0, wnosamp=400, nlayers=60,
                    solution="eclipse", toomuch=10.0, ethresh=1e-4, seed=5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a, b, used = both(P)
    assert_same(a, b)


@pytest.mark.parametrize("xcd_map", ["0", "3"])
def test_ranges_by_xcd_or_in_launch_order(tmp_path, xcd_map):
    """Which workgroup takes which range (xcd_block: one contiguous eighth of the list per XCD, the production
    mapping of k_line_walk_lanes; TRX_XCD_MAP=0: launch order everywhere, 3: by XCD in both walks) is no
    part of a result: every range writes its own partial records."""
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=150_000, wnlow=2500, wnhigh=2900, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=77)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    ref_e = Engine(P.static)
    ref = [ref_e.run(P.atm, P.opts, debug=("e", "tau", "last", "computed")) for _ in range(2)] + [ref_e.run(P.atm, P.opts)]
    ref_e.close()
    os.environ["TRX_XCD_MAP"] = xcd_map
    try:
        e = Engine(P.static)
    finally:
        os.environ.pop("TRX_XCD_MAP", None)
    got = [e.run(P.atm, P.opts, debug=("e", "tau", "last", "computed")) for _ in range(2)] + [e.run(P.atm, P.opts)]
    e.close()
    assert_same(got, ref)
