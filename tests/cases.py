"""Golden cases: inputs + what the compiled reference wrote for them."""
import os

import numpy as np

import oracle_lib as ol
from transit_amd.host import Problem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["eclipse_small", "transit_small", "coadd_thresh", "cloud_scatter", "transit_modm1",
         "highres_fine", "midres_os4", "multi_species", "scat_polar", "qscale_eclipse", "dumps_transit",
         "resample_radius", "resample_transit", "many_isotopes", "saveext_transit",
         "cloud_opa", "cloud_b17", "cloud_f18", "cloud_p19"]


class Golden:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        self.problem = Problem.from_cfg(os.path.join(self.dir, "case.cfg"))
        self.spectrum = ol.read_spectrum(os.path.join(self.dir, "spectrum.dat"))[:, 1]
        _, self.tau = ol.read_rows_dump(os.path.join(self.dir, "tau.dat"), "wavenumber")      # [wn][height]
        _, self.e = ol.read_rows_dump(os.path.join(self.dir, "mol_extion.dat"), "radius")     # [layer][wn]
        _, cia = ol.read_rows_dump(os.path.join(self.dir, "CIA.dat"), "wavenumber")           # [wn][layer]
        self.e_cs = cia.T
        self.swept = (self.e != 0).any(axis=1)          # layers the reference's lazy sweep reached


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


def rel_err(a, b, floor=1e-300):
    a, b = np.asarray(a, float), np.asarray(b, float)
    m = np.abs(b) > floor
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))


def free_port():
    """A TCP port that is free right now (rendezvous of multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
