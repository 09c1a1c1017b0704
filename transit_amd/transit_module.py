"""Drop-in for the reference's SWIG module `transit_module` (transit/src/transit.i:97-105):
the same eight functions with the same Python signatures, so that a retrieval driver
(BART) only changes its import --

    import transit_amd.transit_module as trm      # was: import transit_module as trm
    trm.transit_init(len(argv), argv)              # transit.c:25-74, once: files, samplings, GPU handle
    nwave = trm.get_no_samples()                   # transit.c:77
    wn    = trm.get_waveno_arr(nwave)              # transit.c:82
    trm.set_radius(r0); trm.set_cloudtop(c); trm.set_scattering(flag, logext)   # transit.c:98-116
    spec  = trm.run_transit(profiles, nwave)       # transit.c:118-122: reloadatm + the spectrum path
    trm.free_memory()                              # transit.c:209-228

Like the reference's, the state is module-global (one run per process; one process per GPU).
The line list, Voigt table and CIA tables stay resident on the GPU between run_transit calls;
each call ships the layer arrays (a few KB) and returns the spectrum.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .engine import Engine
from .host import Problem

_problem: Optional[Problem] = None
_engine: Optional[Engine] = None


def _need():
    if _problem is None or _engine is None:
        raise RuntimeError("transit_init() has not been called")
    return _problem, _engine


def transit_init(argc: int, argv: Sequence[str]) -> None:
    """argv as the reference takes it: argv[0] is the program name."""
    global _problem, _engine
    free_memory()
    args: List[str] = [str(a) for a in list(argv)[:argc]][1:]
    P = Problem(args)
    eng = Engine(P.static)
    if P.needs_opacity_build:               # --opacityfile names a file that does not exist yet
        eng.build_opacity_grid(P)            # calcopacity(), then go on in grid mode like the reference
        eng.close()
        eng = Engine(P.static)
    _problem, _engine = P, eng


def get_no_samples() -> int:
    return _need()[0].nwn


def get_waveno_arr(waveno: int) -> np.ndarray:
    wn = _need()[0].wavenumbers()
    out = np.zeros(int(waveno))
    n = min(int(waveno), wn.size)
    out[:n] = wn[:n]
    return out


def set_radius(refradius: float) -> None:
    _need()[0].set_radius(refradius)


def set_cloudtop(cloudtop: float) -> None:
    _need()[0].set_cloudtop(cloudtop)


def set_scattering(flag: int, scattering: float) -> None:
    _need()[0].set_scattering(flag, scattering)


def run_transit(re_input, transit_out_size: int) -> np.ndarray:
    """re_input = [T(nlayer), q_0(nlayer), ..., q_{nmol-1}(nlayer)] (readatm.c:722-784)."""
    P, eng = _need()
    P.reload_atm(np.asarray(re_input, dtype=np.float64))
    spec = eng.run(P.atm, P.opts)["spectrum"]
    out = np.zeros(int(transit_out_size))
    n = min(out.size, spec.size)
    out[:n] = spec[:n]
    return out


def free_memory() -> None:
    global _problem, _engine
    if _engine is not None:
        _engine.close()
    if _problem is not None:
        _problem.close()
    _problem = _engine = None
