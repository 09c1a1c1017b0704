"""The one place where a parity tolerance is wider than the arithmetic's 1e-9, and why.

Mechanism.  The reference's interp_parab (pu/src/numerical.c:182-195) fits its parabola in
ABSOLUTE radius, with u = x[0]/dx:

    a = m/(2 dx^2),  b = (y2 - y1 - (u + 1.5) m)/dx,  c = y0 + u (y2 - 4 y1 + 3 y0 + u m)/2,
    m = y0 + y2 - 2 y1,        value(xr) = xr^2 a + xr b + c

Every ray solution evaluates it AT one of its own nodes (the bottom point of the ray:
eclipse.c:63-66, slantpath.c:60-66), where exact arithmetic returns that node's y.  In doubles
the three terms are each of size ~ |m| (xr/dx)^2 and cancel, so the value carries rounding noise
in steps of

    quantum = ulp(|m| (xr/dx)^2)

however small the input difference that re-rolls the roundings.  For a planet (xr ~ 1e10 cm)
sampled every ~100 km, (xr/dx)^2 ~ 1e6..1e10: a last-bit difference of an extinction value
(the GPU associates a bin's line sum differently from the CPU) comes back as up to ~1e-6 of the
SECOND DIFFERENCE of the extinction -- which, where the extinction jumps between neighbouring
layers or the atmosphere has 3-4 layers, is the size of the extinction itself.  The reference
built with and without -ffast-math differs from itself in the same way.
tests/test_tolerance_mechanism.py demonstrates this on the CPU with trxo_parab3.

What the helpers allow: |d tau| <= STRICT * (largest tau of the ray)  +  NQ quanta of every
parabola value that enters that optical depth, each times the path length it is weighted with.
A difference that is large but NOT of this form -- at a height whose parabola is smooth, or many
quanta wide -- fails.
"""
import numpy as np

STRICT = 1e-9      # the arithmetic's own tolerance (fp64, sums associated differently)
NQ = 4.0           # rounding steps allowed per parabola value (3 cancelling terms, 2 results compared)


def parab_quantum(x_lo, x_hi, y0, y1, y2, xr):
    """Size of one rounding step of interp_parab(x=[x_lo, x_hi, ...], y=[y0, y1, y2], xr)."""
    dx = x_hi - x_lo
    m = np.abs(y0 + y2 - 2 * y1)
    return np.spacing(m * (xr / dx) ** 2 + np.abs(y0) + np.abs(y1) + np.abs(y2))


def total_extinction(dbg):
    """tau.c:231-232 from a debug run's arrays ([layer][wn])."""
    return dbg["e"] + dbg["e_scat"] + dbg["e_cloud"] + dbg["e_cs"]


def layer_quanta(radius, y):
    """[layer][wn]: the quantum of the parabola value a ray bottoming in that layer uses
    (eclipse.c:63-66 / slantpath.c:60-66: nodes rs..rs+2, or rs-1..rs+1 when only two remain)."""
    nr = len(radius)
    q = np.zeros_like(y)
    for rs in range(nr - 1):
        j = rs if nr - rs >= 3 else rs - 1
        if j < 0:
            continue
        q[rs] = parab_quantum(radius[j], radius[j + 1], y[j], y[j + 1], y[j + 2], radius[rs])
    return q


def tau_allowance(problem, dbg, parts=False):
    """[wn][height] absolute allowance on the optical depth of a debug run (`dbg` must hold
    e, e_scat, e_cloud, e_cs and tau), heights top first as in trx_debug.tau.  parts: the
    arithmetic's share and the parabola quanta separately."""
    rad = problem.layer_arrays()["radius"]
    fct = float(problem.atm.rad_fct)
    nr = len(rad)
    q = layer_quanta(rad, total_extinction(dbg))                       # [layer][wn]
    span = np.empty(nr)                                                 # path a node's value is weighted with
    span[1:-1] = rad[2:] - rad[:-2]
    span[0], span[-1] = rad[1] - rad[0], rad[-1] - rad[-2]
    allow = np.zeros_like(dbg["tau"])
    if problem.opts.solution == 0:
        # eclipse: the bottom-point values stay in er (eclipse.c:65-66), so a ray carries the
        # noise of every layer above its bottom
        acc = np.zeros(q.shape[1])
        for ri in range(1, nr):
            rs = nr - 1 - ri
            acc = acc + q[rs] * span[rs]
            allow[:, ri] = NQ * fct * acc
    else:
        # transit: only the ray's own bottom point is a parabola value (slantpath.c:60-66,
        # restored afterwards), weighted with the chord to the next layer, both ways
        for ri in range(1, nr):
            rs = nr - 1 - ri
            chord = np.sqrt(max(rad[rs + 1] ** 2 - rad[rs] ** 2, 0.0))
            allow[:, ri] = NQ * fct * q[rs] * 2 * chord
    scale = np.abs(dbg["tau"]).max(axis=1, keepdims=True)
    if parts:
        return STRICT * scale, allow
    return STRICT * scale + allow


def assert_tau_close(problem, got, ref, note=None):
    """got["tau"] against ref["tau"] (ref a debug run with the arrays tau_allowance needs).
    Returns the rays that CAN carry parabola noise into the spectrum: those for which, at some
    height that enters the spectrum (down to the ray's toomuch crossing, ref["last"]), the quanta
    allowed there exceed the arithmetic's own 1e-9 of THAT height's optical depth.  (By the ray's
    largest optical depth instead -- 1e4..1e6 at the bottom of an opaque ray -- a three-layer
    atmosphere whose extinction jumps 1e7 between layers looked noise-free while its middle
    height, the one the spectrum is made of, was 1e-6 off inside the allowance.)"""
    diff = np.abs(got["tau"] - ref["tau"])
    strict, quanta = tau_allowance(problem, ref, parts=True)
    allow = strict + quanta
    bad = diff > allow
    if bad.any():
        w, ri = np.argwhere(bad)[0]
        raise AssertionError("tau differs by %.3e at wn %d height %d: allowed %.3e (%.3e strict + %.1f quanta); %s"
                             % (diff[w, ri], w, ri, allow[w, ri], STRICT * np.abs(ref["tau"][w]).max(), NQ, note))
    nh = ref["tau"].shape[1]
    used = np.arange(nh)[None, :] <= np.asarray(ref["last"])[:, None] if "last" in ref else np.ones_like(bad)
    return (used & (quanta > STRICT * np.abs(ref["tau"]))).any(axis=1)


DEBUG_KEYS = ("e", "e_cs", "tau", "last", "intens", "computed", "e_scat", "e_cloud")
