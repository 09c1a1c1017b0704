"""A shard sizes its walk frames for the lines that can reach IT (prep_layers: psmax of a windowed
handle).  Doppler widths grow with the wavenumber, so on a band that spans a factor of several the
widest profile of the whole list is much wider than the widest one near a low-wavenumber shard.  The
frame is only a bound -- which bins a group may touch -- so the shard's numbers must not move: against
the same shard with frames sized for the whole list (TRX_SHARD_FRAMES=0), and stitched against the
unsharded run."""
import os

import numpy as np
import pytest

from cases import rel_err
from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem
from transit_amd.shard import all_bounds

pytestmark = pytest.mark.gpu

KEYS = ("e", "tau", "last", "computed")


def frames_of(msgs):
    for m in msgs:
        if "walk frame (bins) per layer" in m:
            return [int(x) for x in m.split(":")[-1].split()]
    return None


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_shard_frames_do_not_move_the_numbers(tmp_path, solution):
    d = str(tmp_path / "w")
    synth.make_case(d, nlines=300_000, wnlow=2000, wnhigh=14000, wndelt=1.0, wnosamp=2160, nlayers=60,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=21, ncia=1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    try:
        e = Engine(P.static)
        full = e.run(P.atm, P.opts, debug=KEYS)
        e.close()
        f_tight = frames_of(msgs)
        os.environ["TRX_SHARD_FRAMES"] = "0"
        try:
            e = Engine(P.static)
        finally:
            os.environ.pop("TRX_SHARD_FRAMES", None)
        del msgs[:]
        loose = e.run(P.atm, P.opts, debug=KEYS)
        e.close()
        f_full = frames_of(msgs)
        # the unsharded run: frames sized for the indices its lines can take against frames sized for
        # the isotopes' whole wavenumber range
        assert all(x <= y or y == 0 for x, y in zip(f_tight, f_full)), (f_tight, f_full)
        assert np.array_equal(full["last"], loose["last"])
        assert np.allclose(full["spectrum"], loose["spectrum"], rtol=1e-11, atol=0)
        if all((x == 0) == (y == 0) for x, y in zip(f_tight, f_full)):
            assert np.array_equal(full["spectrum"], loose["spectrum"])
        narrower = 0
        parts, lasts = [], []
        for k, (lo, hi) in enumerate(all_bounds(P.nwn, 6)):
            P.set_shard(lo, hi)
            try:
                del msgs[:]
                a = Engine(P.static)
                ra = [a.run(P.atm, P.opts, debug=KEYS), a.run(P.atm, P.opts, debug=KEYS), a.run(P.atm, P.opts)]     # unhinted, hinted, production
                a.close()
                f_shard = frames_of(msgs)
                os.environ["TRX_SHARD_FRAMES"] = "0"
                try:
                    b = Engine(P.static)
                finally:
                    os.environ.pop("TRX_SHARD_FRAMES", None)
                del msgs[:]
                rb = [b.run(P.atm, P.opts, debug=KEYS), b.run(P.atm, P.opts, debug=KEYS), b.run(P.atm, P.opts)]
                b.close()
                f_list = frames_of(msgs)
            finally:
                P.set_shard(0, P.nwn)
            assert f_list == f_full, k                    # (sized for the whole list: what the unsharded run uses)
            assert all(x <= y or y == 0 for x, y in zip(f_shard, f_list)), (k, f_shard, f_list)
            narrower += f_shard != f_list
            same_kind = all((x == 0) == (y == 0) for x, y in zip(f_shard, f_list))
            for x, y in zip(ra, rb):
                if "last" in x:
                    assert np.array_equal(x["last"], y["last"]), k
                    sw = x["computed"].astype(bool) & y["computed"].astype(bool)
                    assert rel_err(x["e"][sw], y["e"][sw]) < 1e-12, k
                if same_kind:
                    # walks on both sides: one owner per bin, line order -- the same bits whatever the frame
                    assert np.array_equal(x["spectrum"], y["spectrum"]), k
                else:
                    assert rel_err(x["spectrum"], y["spectrum"]) < 1e-11, k
            parts.append(ra[0]["spectrum"]); lasts.append(ra[0]["last"])
        assert narrower >= 2, "no shard got narrower frames: the case does not test what it should"
        assert np.array_equal(np.concatenate(lasts), full["last"])
        assert rel_err(np.concatenate(parts), full["spectrum"]) < (1e-9 if solution == "transit" else 1e-11)
    finally:
        engine.set_log(None)
