"""Test-side binding of the CPU oracle (oracle/libtrx_oracle.so).

Lives under tests/ on purpose: the product package never imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from transit_amd import _abi
from transit_amd.engine import CEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


def oracle_library():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libtrx_oracle.so")
        src = os.path.join(ORACLE_DIR, "trx_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
        lib = C.CDLL(so)
        _abi.bind_engine_api(lib, "trxo_")
        lib.trxo_voigt_profile.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _abi.c_float_p, C.c_int]
        lib.trxo_voigt_profile.restype = C.c_int
        lib.trxo_parab3.argtypes = [_abi.c_double_p, _abi.c_double_p, C.c_double]
        lib.trxo_parab3.restype = C.c_double
        lib.trxo_simpson.argtypes = [_abi.c_double_p, _abi.c_double_p, C.c_int]
        lib.trxo_simpson.restype = C.c_double
        lib.trxo_tau_slant.argtypes = [_abi.c_double_p, C.c_long, C.c_double, _abi.c_double_p]
        lib.trxo_tau_slant.restype = C.c_double
        lib.trxo_modulation.argtypes = [_abi.c_double_p, C.c_long, C.c_double, _abi.c_double_p, C.c_long, C.c_double, C.c_double, C.c_int]
        lib.trxo_modulation.restype = C.c_double
        lib.trxo_nearest.argtypes = [_abi.c_double_p, C.c_double, C.c_int, C.c_int]
        lib.trxo_nearest.restype = C.c_int
        lib.trxo_spline_init.argtypes = [_abi.c_double_p, _abi.c_double_p, _abi.c_double_p, C.c_long]
        lib.trxo_spline_init.restype = None
        lib.trxo_spline_eval.argtypes = [_abi.c_double_p, C.c_long, _abi.c_double_p, _abi.c_double_p, C.c_double]
        lib.trxo_spline_eval.restype = C.c_double
        _lib = lib
    return _lib


class OracleEngine(CEngine):
    def __init__(self, static):
        super().__init__(oracle_library(), "trxo_", static)


def ref_pu_library():
    p = os.path.join(ORACLE_DIR, "_ref", "libpu_ref.so")
    if not os.path.exists(p):
        return None
    lib = C.CDLL(p)
    lib.voigtn.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(_abi.c_float_p), C.c_double, C.c_int]
    lib.voigtn.restype = C.c_int
    return lib


# ---- parsers of the reference's text outputs ---------------------------------
def read_spectrum(path):
    return np.loadtxt(path, comments="#")           # [nwn, 2]: wavelength um, value


def read_rows_dump(path, key):
    """Blocks 'key: x' followed by one line of numbers (tau.dat, CIA.dat,
    mol_extion.dat, *_extion.dat; writers tau.c:360-518)."""
    heads, rows = [], []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith(key):
            heads.append(float(ln.split(":")[1]))
            rows.append(np.array(lines[i + 1].split(), dtype=float))
            i += 2
        else:
            i += 1
    return np.array(heads), np.vstack(rows)
