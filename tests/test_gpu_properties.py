"""Size-independent properties of the HIP path, also at the bench's full size
(BASELINE configs[1]: 2501 wavenumbers x 100 layers x 10^6 lines), where the
CPU oracle would take a minute per case."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from cases import golden, rel_err
from tolerances import DEBUG_KEYS, assert_tau_close
from transit_amd import _abi, synth
from transit_amd.engine import Engine, EngineError
from transit_amd.host import Problem
from transit_amd.shard import all_bounds

pytestmark = pytest.mark.gpu


def clone_static(st, **arrays):
    """Copy of a trx_static with some line arrays replaced (arrays kept alive on the copy)."""
    st2 = _abi.TrxStatic.from_buffer_copy(st)
    st2._keep = arrays
    for name, arr in arrays.items():
        ptr_t = dict(_abi.TrxStatic._fields_)[name]
        setattr(st2, name, arr.ctypes.data_as(ptr_t))
    return st2


def line_array(st, name, dtype=np.float64):
    return np.ctypeslib.as_array(getattr(st, name), shape=(st.nlines,)).astype(dtype).copy()


@pytest.fixture(scope="module")
def full_size(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("c2"))
    synth.make_case(d, nlines=1_000_000, wnlow=2500, wnhigh=5000, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=1234)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    eng = Engine(P.static)
    out = eng.run(P.atm, P.opts, debug=True)
    yield P, eng, out
    eng.close()


def test_full_size_optical_depth_is_monotonic_and_cut_consistent(full_size):
    P, eng, out = full_size
    tau, last = out["tau"], out["last"]
    assert np.all(np.isfinite(out["spectrum"])) and np.all(out["spectrum"] > 0)
    nr = P.nlayer
    for w in range(0, P.nwn, 37):
        t = tau[w, : last[w] + 1]
        assert np.all(np.diff(t) >= -1e-12 * t[-1])             # deeper never gets more transparent
        assert np.all(tau[w, last[w] + 1:] == 0)                # nothing computed below the cut
    crossed = tau[np.arange(P.nwn), last] > P.opts.toomuch
    assert np.all(crossed | (last == nr - 1))
    before = tau[np.arange(P.nwn), np.maximum(last - 1, 0)]
    assert np.all((before <= P.opts.toomuch) | (last == 0))


def test_full_size_extinction_is_linear_in_gf(full_size):
    """Scaling every gf by 2 scales every line strength, the layer maximum and the
    threshold by exactly 2: the molecular extinction must double bit for bit."""
    P, eng, out = full_size
    gf2 = line_array(P.static, "gf") * 2.0
    e2 = Engine(clone_static(P.static, gf=gf2))
    opts = P.opts
    opts.eager = 1
    try:
        a = eng.run(P.atm, opts, debug=True)["e"]
        b = e2.run(P.atm, opts, debug=True)["e"]
    finally:
        opts.eager = 0
        e2.close()
    assert a.max() > 0
    assert np.array_equal(b, 2.0 * a)


def test_full_size_shards_stitch(full_size):
    P, eng, out = full_size
    parts = []
    try:
        for lo, hi in all_bounds(P.nwn, 4):
            P.set_shard(lo, hi)
            e = Engine(P.static)
            parts.append(e.run(P.atm, P.opts)["spectrum"])
            e.close()
    finally:
        P.set_shard(0, P.nwn)
    # tile boundaries move with the shard origin, so the per-bin sums (~400 terms) are re-associated
    assert rel_err(np.concatenate(parts), out["spectrum"]) < 1e-10


def test_full_size_against_oracle_on_a_line_subset(full_size):
    """Oracle parity at the full grid and layer count with 2 % of the lines (what
    the CPU finishes in seconds); same generator, same atmosphere."""
    P, _, _ = full_size
    d = os.path.join(os.path.dirname(P.cwd), "c2_sub")
    synth.make_case(d, nlines=20_000, wnlow=2500, wnhigh=5000, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=1234)
    Q = Problem.from_cfg(os.path.join(d, "case.cfg"))
    hip = Engine(Q.static)
    got = hip.run(Q.atm, Q.opts, debug=True)
    hip.close()
    ora = ol.OracleEngine(Q.static)
    ref = ora.run(Q.atm, Q.opts, debug=True)
    ora.close()
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert rel_err(got["tau"], ref["tau"]) < 1e-9


# ---- edge cases on the small fixtures ---------------------------------------------
def _both(static, P):
    hip = Engine(static)
    got = hip.run(P.atm, P.opts, debug=True)
    hip.close()
    ora = ol.OracleEngine(static)
    ref = ora.run(P.atm, P.opts, debug=DEBUG_KEYS)
    ora.close()
    return got, ref


def test_empty_line_list_is_cia_only():
    P = golden("eclipse_small").problem
    st = _abi.TrxStatic.from_buffer_copy(P.static)
    st.nlines = 0
    got, ref = _both(st, P)
    assert np.all(got["e"] == 0)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-12


def test_lines_outside_the_band_contribute_nothing():
    P = golden("eclipse_small").problem
    wl = line_array(P.static, "wl_um")
    far = np.sort(1e4 / np.linspace(2700.0, 2800.0, wl.size))        # ascending wavelength, out of range
    iso = line_array(P.static, "isoid", np.int16)
    # keep TLI order: isotope blocks ascending, wavelength ascending inside a block
    for b in np.unique(iso):
        m = iso == b
        wl[m] = far[: m.sum()]
    st = clone_static(P.static, wl_um=wl)
    got, ref = _both(st, P)
    assert np.all(got["e"] == 0) and np.all(ref["e"] == 0)
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-12


def _dirty_the_device_pool(P):
    """A run with lines and dumps leaves non-zero extinction in freed device memory."""
    hip = Engine(P.static)
    out = hip.run(P.atm, P.opts, debug=True)
    hip.close()
    assert (out["e"] != 0).any()


@pytest.mark.parametrize("kind", ["no_lines", "lines_outside_band"])
def test_no_inrange_line_without_debug_dumps(kind):
    """No kernel writes the molecular extinction when no line is in range; the plain
    run (no dumps requested, which used to be what zeroed it) must still read zeros."""
    P = golden("eclipse_small").problem
    _dirty_the_device_pool(P)
    if kind == "no_lines":
        st = _abi.TrxStatic.from_buffer_copy(P.static)
        st.nlines = 0
    else:
        wl = line_array(P.static, "wl_um")
        far = np.sort(1e4 / np.linspace(2700.0, 2800.0, wl.size))
        iso = line_array(P.static, "isoid", np.int16)
        for b in np.unique(iso):
            m = iso == b
            wl[m] = far[: m.sum()]
        st = clone_static(P.static, wl_um=wl)
    hip = Engine(st)
    first = hip.run(P.atm, P.opts)["spectrum"]
    second = hip.run(P.atm, P.opts)["spectrum"]          # hinted plan, same answer
    hip.close()
    ora = ol.OracleEngine(st)
    ref = ora.run(P.atm, P.opts)["spectrum"]
    ora.close()
    assert rel_err(first, ref) < 1e-12
    assert np.array_equal(first, second)


def test_single_line_on_a_grid_point_and_at_the_band_edges():
    """Three hand-placed lines: exactly on a coarse grid point, on the first and on
    the last wavenumber of the band (clamped windows, extinction.c:493-496)."""
    P = golden("eclipse_small").problem
    st0 = P.static
    wn = np.array([2560.0, 2530.0, 2500.0])                          # descending = ascending wavelength
    st = clone_static(st0, wl_um=1e4 / wn, elow=np.array([100.0, 500.0, 1500.0]),
                      gf=np.array([1e-6, 3e-6, 1e-5]), isoid=np.zeros(3, dtype=np.int16))
    st.nlines = 3
    got, ref = _both(st, P)
    assert (ref["e"] != 0).any()
    sw = ref["computed"].astype(bool)
    assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-10
    assert np.array_equal(got["e"][sw] == 0, ref["e"][sw] == 0)
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-10


def test_unsorted_line_list_is_rejected():
    from transit_amd.engine import EngineError
    P = golden("eclipse_small").problem
    wl = line_array(P.static, "wl_um")
    wl[[10, 11]] = wl[[11, 10]] if wl[10] != wl[11] else (wl[10] * 1.001, wl[10])
    wl[10], wl[11] = max(wl[10], wl[11]), min(wl[10], wl[11])        # descending pair inside a block
    with pytest.raises(EngineError) as ei:
        Engine(clone_static(P.static, wl_um=wl))
    assert ei.value.code == -7


def test_bad_arguments_return_codes():
    from transit_amd.engine import EngineError
    P = golden("eclipse_small").problem
    st = _abi.TrxStatic.from_buffer_copy(P.static)
    st.abi_version = 99
    with pytest.raises(EngineError) as ei:
        Engine(st)
    assert ei.value.code == -1
    eng = Engine(P.static)
    o = _abi.TrxOpts.from_buffer_copy(P.opts)
    o.ethresh = 0.0
    with pytest.raises(EngineError):
        eng.run(P.atm, o)
    o = _abi.TrxOpts.from_buffer_copy(P.opts)
    o.solution = 7
    with pytest.raises(EngineError):
        eng.run(P.atm, o)
    # CIA table range: a layer hotter than the table allows -> TRX_E_RANGE, not a crash
    arr = P.layer_arrays()
    hot = arr["temp"].copy(); hot[0] = 9000.0
    a = _abi.TrxAtm.from_buffer_copy(P.atm)
    a.temp = hot.ctypes.data_as(_abi.c_double_p)
    with pytest.raises(EngineError) as ei:
        eng.run(a, P.opts)
    assert ei.value.code == -5
    # the handle is still usable afterwards and nothing of the failed run leaks into the next
    r = eng.run(P.atm, P.opts)
    eng.close()
    fresh = Engine(P.static)
    ref = fresh.run(P.atm, P.opts)
    fresh.close()
    assert np.array_equal(r["spectrum"], ref["spectrum"])


def test_many_isotopes_against_oracle(tmp_path):
    """Ten isotopes in two databases (more than the per-layer scalars staged in LDS)."""
    names = tuple("i%d" % k for k in range(7))
    db1 = synth.synth_linedb(2100, 2500, 2530, seed=3, name="A", molname="H2O", iso_names=names,
                             iso_masses=tuple(18.0 + 0.3 * k for k in range(7)),
                             iso_ratios=tuple(0.5 / (k + 1) for k in range(7)),
                             iso_split=(0.3, 0.2, 0.15, 0.1, 0.1, 0.1, 0.05), z_scale=170.0)
    db2 = synth.synth_linedb(900, 2500, 2530, seed=4, name="B", molname="CO2", iso_names=("626", "636", "628"),
                             iso_masses=(43.98983, 44.993185, 45.994076), iso_ratios=(0.984204, 0.011057, 0.003947),
                             iso_split=(0.6, 0.3, 0.1), z_scale=286.0)
    d = str(tmp_path / "iso10")
    synth.make_case(d, wnlow=2500, wnhigh=2530, nlayers=18, solution="eclipse", ethresh=1e-5, toomuch=9.0,
                    dbs=[db1, db2])
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.static.niso == 10
    got, ref = _both(P.static, P)
    sw = ref["computed"].astype(bool)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-9
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9


@pytest.mark.parametrize("solution,scales", [("eclipse", (1.0, 0.02, 1.0, 30.0, 0.02)),
                                             ("transit", (1.0, 0.3, 1.0, 3.0, 0.3))])
def test_depth_hint_is_only_a_hint(tmp_path, solution, scales):
    """A handle plans its steps from the depth the previous spectrum reached.  When the next
    atmosphere is more transparent (rays go deeper) or more opaque (they stop earlier) the
    result must be the one a fresh handle gives."""
    d = str(tmp_path / "hint")
    synth.make_case(d, nlines=30_000, wnlow=2500, wnhigh=2700, wndelt=1.0, wnosamp=2160, nlayers=80,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=5)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    a = P.atm
    dens = np.ctypeslib.as_array(a.density, shape=(P.static.nmol * P.nlayer,))
    base = dens.copy()

    def fresh():
        e = Engine(P.static)
        out = e.run(P.atm, P.opts, debug=("last", "tau"))
        e.close()
        return out

    eng = Engine(P.static)
    try:
        seen = []
        for scale in scales:
            dens[:] = base * scale
            got = eng.run(P.atm, P.opts, debug=("last", "tau"))
            ref = fresh()
            seen.append(int(ref["last"].max()))
            assert np.array_equal(got["last"], ref["last"]), scale
            assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-13, scale
            assert rel_err(got["tau"], ref["tau"]) < 1e-13, scale
        assert seen[1] > seen[0] > seen[3]          # the sequence really moved the depth both ways
    finally:
        dens[:] = base
        eng.close()


def test_messages_go_to_the_callback():
    """trx_set_log: the library itself never prints; errors and the create summary arrive
    at the callback with the reference's verbosity levels (flags_tr.h:107-111)."""
    from transit_amd import engine
    P = golden("eclipse_small").problem
    seen = []
    engine.set_log(lambda lvl, msg: seen.append((lvl, msg)), 3)
    try:
        e = Engine(P.static)
        opts = P.opts
        old = opts.ethresh
        opts.ethresh = -1.0
        with pytest.raises(engine.EngineError):
            e.run(P.atm, opts)
        opts.ethresh = old
        e.close()
    finally:
        engine.set_log(None)
    assert any(lvl == 3 and "co-added groups" in msg for lvl, msg in seen)
    assert any(lvl == 1 and "ethresh" in msg for lvl, msg in seen)


@pytest.mark.parametrize("cloud", ["ext,1e-9,-3.0,0.5", "opa,1e-3,-3.0,0.5", "B17,1e-7,-3.0,0.5,-1.5",
                                   "F18,1e-7,-3.0,0.5,2.0,0.8,1e-5", "P19,1e-9,-3.0,0.5,-2.0,1e-22,3000.0"])
@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_cloud_models_against_oracle(tmp_path, cloud, solution):
    """The five cloud parametrisations of extinction.c:630-693 (--cloud type,ext,top,bot,...) in both
    geometries against the CPU restatement.  (The reference itself pins them too since round 4: goldens
    cloud_opa / cloud_b17 / cloud_f18 / cloud_p19 from its build with tau.c's mean_dens[] zero-initialised,
    DESIGN.md section 6 -- tests/test_oracle_golden.py, tests/test_gpu_parity.py.)"""
    d = str(tmp_path / "cl")
    synth.make_case(d, nlines=1500, wnlow=3000, wnhigh=3030, nlayers=30, solution=solution, seed=7,
                    extra={"cloud": cloud})
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    got, ref = _both(P.static, P)
    assert np.all(np.isfinite(ref["spectrum"]))
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert_tau_close(P, got, ref)


def test_two_handles_on_two_threads():
    """Handles share nothing: two of them driven from two host threads at once (ctypes drops the
    GIL during trx_run) give what each gives alone."""
    import threading
    P1, P2 = golden("eclipse_small").problem, golden("coadd_thresh").problem
    refs = []
    for P in (P1, P2):
        e = Engine(P.static)
        refs.append(e.run(P.atm, P.opts)["spectrum"])
        e.close()
    engs = [Engine(P1.static), Engine(P2.static)]
    outs = [[], []]

    def work(k, P):
        for _ in range(40):
            outs[k].append(engs[k].run(P.atm, P.opts)["spectrum"])

    th = [threading.Thread(target=work, args=(0, P1)), threading.Thread(target=work, args=(1, P2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engs:
        e.close()
    for k in range(2):
        assert len(outs[k]) == 40
        for o in outs[k]:
            assert np.array_equal(o, refs[k])


def test_every_walk_frame_and_lane_form_against_oracle(tmp_path):
    """A demo-shaped atmosphere down to 100 bar meets every form of the line sweep: frames of 2, 4,
    8 and 16 bins and the two-kernel form below them.  Swept eagerly, once in steps as large as
    the plan allows (one lane per layer where a wide frame has more than 32 layers) and once in
    steps of 16 layers (two lanes per layer in the frames of 8+ bins): the extinction of every
    layer against the oracle's, and the two runs against each other bit for bit -- the order of a
    bin's sum does not depend on the steps."""
    from transit_amd import engine
    d = str(tmp_path / "case")
    synth.make_case(d, nlines=20000, wnlow=2500, wnhigh=2560, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=91)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    ora = ol.OracleEngine(P.static)
    P.opts.eager = 1
    ref = ora.run(P.atm, P.opts, debug=("e",))
    ora.close()
    seen, runs = [], []
    engine.set_log(lambda lvl, msg: seen.append(msg), 5)

    def frames_of():
        fr = [m for m in seen if "walk frame (bins) per layer" in m]
        assert fr, "no frame report in the debug log"
        return [int(t) for t in fr[0].split(":")[-1].split()]          # top layer first

    try:
        eng = Engine(P.static)
        P.opts.layer_chunk = 0
        runs.append(eng.run(P.atm, P.opts, debug=("e",)))
        frames = frames_of()
        # a step size whose steps (equal parts from the top) include one that holds 8-bin frames and nothing wider:
        # that step is k_line_walk_lanes<8>'s (a dense list); the 16-bin frames get theirs in any step of <= 32 layers
        def step_frames(c):                        # the step plan of trx_run for layer_chunk = c (plan_step: equal parts of what is left)
            nwalk = frames.index(0) if 0 in frames else len(frames)
            pos, out = 0, []
            while pos < nwalk:
                left = nwalk - pos
                nc = -(-left // -(-left // c))
                if pos == 0:
                    nc = max(nc, 3)
                out.append(max(frames[pos:pos + nc]))
                pos += nc
            return out
        chunk8 = next((c for c in range(3, 17) if 8 in step_frames(c)), None)
        assert chunk8 is not None, frames
        for chunk in (16, chunk8):
            P.opts.layer_chunk = chunk
            runs.append(eng.run(P.atm, P.opts, debug=("e",)))
        eng.close()
    finally:
        engine.set_log(None)
        P.opts.layer_chunk = 0
        P.opts.eager = 0
    kinds = set(frames)
    assert kinds == {0, 2, 4, 8, 16}, kinds
    # the 16-layer steps and the steps of the chosen size put the wide frames through k_line_walk_lanes: both of
    # its instantiations are what is compared with the oracle below
    assert any("walk: lanes = lines" in m and "8-bin frames" in m for m in seen), "k_line_walk_lanes<8> did not run"
    assert any("walk: lanes = lines" in m and "16-bin frames" in m for m in seen), "k_line_walk_lanes<16> did not run"
    for r in runs:
        assert rel_err(r["e"], ref["e"]) < 1e-9
        assert rel_err(r["spectrum"], ref["spectrum"]) < 1e-9
    assert np.array_equal(runs[0]["e"], runs[1]["e"]) and np.array_equal(runs[0]["e"], runs[2]["e"])


def test_wide_frames_without_the_row_copy_give_the_same_bits(tmp_path, monkeypatch):
    """Frames of 4+ bins read a row copy of the Voigt table (k_line_walk, `tabW`); when that copy
    would pass 4 GB a handle has none and those frames run their per-bin form.  Same spectrum, bit
    for bit, and the same extinction in every layer (the deep layers are the wide frames)."""
    d = str(tmp_path / "case")
    synth.make_case(d, nlines=30000, wnlow=2500, wnhigh=2600, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution="eclipse", toomuch=10.0, ethresh=1e-50, seed=77)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    outs = []
    for no_copy in (False, True):
        if no_copy:
            monkeypatch.setenv("TRX_NO_ROW_COPY", "1")
        eng = Engine(P.static)
        P.opts.eager = 1                                    # every layer, also below where the rays stop
        outs.append(eng.run(P.atm, P.opts, debug=("e",)))
        P.opts.eager = 0
        assert eng.stats()["walk_steps"] > 0
        eng.close()
    assert np.array_equal(outs[0]["spectrum"], outs[1]["spectrum"])
    assert np.array_equal(outs[0]["e"], outs[1]["e"])
    assert np.count_nonzero(outs[0]["e"][0]) > 0           # the bottom layer (widest profiles) was swept by something


@pytest.mark.parametrize("grid", [dict(ndop=256, nlor=7), dict(ndop=2, nlor=2), dict(ndop=17, nlor=120, dmin=5e-4, dmax=0.5, lmin=1e-3, lmax=2.0)])
def test_other_voigt_grids_against_oracle(tmp_path, grid):
    """--ndop/--nlor/--dmin/... (argum.c:220-237): the largest Doppler grid the kernels stage in
    LDS, the smallest legal one (every line falls on its edges), and an asymmetric one."""
    d = str(tmp_path / "vg")
    synth.make_case(d, nlines=2500, wnlow=2500, wnhigh=2530, nlayers=25, solution="eclipse", seed=12,
                    ethresh=1e-7, extra={k: str(v) for k, v in grid.items()})
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.static.ndop == grid["ndop"] and P.static.nlor == grid["nlor"]
    got, ref = _both(P.static, P)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    sw = got["computed"].astype(bool) & ref["computed"].astype(bool)
    assert rel_err(got["e"][sw], ref["e"][sw]) < 1e-9


def test_sixty_five_isotopes_of_one_molecule(tmp_path):
    """One database with 65 isotopes (round 2's bound was 64; the bound now, kMaxIso = 256, and
    130 isotopes in 8 databases: tests/test_gpu_isotopes.py)."""
    def db(n):
        names = tuple("i%02d" % k for k in range(n))
        return synth.synth_linedb(64 * n, 2500, 2520, seed=3, name="many", molname="H2O", iso_names=names,
                                  iso_masses=tuple(18.0 + 0.05 * k for k in range(n)),
                                  iso_ratios=tuple(0.5 / (k + 1) for k in range(n)),
                                  iso_split=tuple(1.0 / n for _ in range(n)), z_scale=170.0)
    d = str(tmp_path / "i64")
    synth.make_case(d, wnlow=2500, wnhigh=2520, nlayers=12, solution="eclipse", dbs=[db(65)], ethresh=1e-6)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.static.niso == 65
    got, ref = _both(P.static, P)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9


def test_large_grid_modulation_kernel_against_oracle(tmp_path):
    """The transit counterpart: above 65 536 wavenumbers the modulation integral runs one lane per
    wavenumber (k_modulation_rows); spectrum and optical depth against the CPU restatement."""
    d = str(tmp_path / "wide_t")
    synth.make_case(d, nlines=60, wnlow=2500, wnhigh=2507, wndelt=1e-4, wnosamp=1, nlayers=9, solution="transit", seed=6,
                    toomuch=10.0, nwidth=2.0)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn > 65536
    got, ref = _both(P.static, P)
    assert np.array_equal(got["last"], ref["last"])
    assert_tau_close(P, got, ref)
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-8


def test_large_shard_emission_kernel_against_oracle(tmp_path):
    """A 70 001-point grid with few layers: the small-shard variants of the optical-depth and
    start-up kernels switch off above 65 536 wavenumbers; spectrum and per-angle intensities
    against the CPU restatement."""
    d = str(tmp_path / "wide")
    synth.make_case(d, nlines=60, wnlow=2500, wnhigh=2507, wndelt=1e-4, wnosamp=1, nlayers=8, solution="eclipse", seed=5,
                    toomuch=10.0, nwidth=2.0)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    assert P.nwn > 65536
    got, ref = _both(P.static, P)
    assert np.array_equal(got["last"], ref["last"])
    assert rel_err(got["spectrum"], ref["spectrum"]) < 1e-9
    assert rel_err(got["intens"], ref["intens"]) < 1e-9


@pytest.mark.parametrize("osamp,wndelt", [(1, 0.004), (2160, 1.0)])
def test_threaded_create_gives_the_single_thread_groups(tmp_path, osamp, wndelt):
    """trx_create groups the list on several host threads (pieces cut where the co-add chain
    provably restarts); groups, co-add count and spectrum must be the single-threaded ones --
    on a list dense enough to co-add (0.4 lines per fine cell) and on a sparse one."""
    d = str(tmp_path / "thr")
    synth.make_case(d, nlines=400_000, wnlow=2500, wnhigh=2500 + (4000 if osamp == 1 else 2000), wndelt=wndelt, wnosamp=osamp,
                    nlayers=12, solution="eclipse", seed=9, nwidth=2.0)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    res = []
    for nth in ("1", "7"):
        os.environ["TRX_CREATE_THREADS"] = nth
        try:
            e = Engine(P.static)
        finally:
            os.environ.pop("TRX_CREATE_THREADS", None)
        out = e.run(P.atm, P.opts)
        st = e.stats()
        e.close()
        res.append((out["spectrum"], st["ngroups"], st["nadd"], st["nlines_inrange"]))
    assert res[0][1:] == res[1][1:]
    assert np.array_equal(res[0][0], res[1][0])
    if osamp == 1:
        assert res[0][2] > 10_000          # the dense list does co-add
