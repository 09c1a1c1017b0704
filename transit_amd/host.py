"""Python face of the C++ host side (include/transit_host.h).

`Problem.from_cfg("run.cfg")` does what the reference's transit_init() does
before the spectrum path (transit/src/transit.c:25-74): options, samplings,
atmosphere, TLI, CIA -- and hands out the plain structs that the engine eats.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _abi
from .build import lib_path

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = lib_path("libtransit_host.so")
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = C.CDLL(path)
        lib.trh_load.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        lib.trh_load.restype = C.c_int
        lib.trh_free.argtypes = [C.c_void_p]
        lib.trh_free.restype = None
        for name, res in (("trh_static", C.POINTER(_abi.TrxStatic)), ("trh_atm", C.POINTER(_abi.TrxAtm)),
                          ("trh_opts", C.POINTER(_abi.TrxOpts))):
            getattr(lib, name).argtypes = [C.c_void_p]
            getattr(lib, name).restype = res
        lib.trh_nwn.argtypes = [C.c_void_p]
        lib.trh_nwn.restype = C.c_int64
        lib.trh_wavenumbers.argtypes = [C.c_void_p, _abi.c_double_p]
        lib.trh_wavenumbers.restype = None
        lib.trh_set_shard.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        lib.trh_set_shard.restype = None
        lib.trh_reload_atm.argtypes = [C.c_void_p, _abi.c_double_p, C.c_int]
        lib.trh_reload_atm.restype = C.c_int
        lib.trh_set_radius.argtypes = [C.c_void_p, C.c_double]
        lib.trh_set_cloudtop.argtypes = [C.c_void_p, C.c_double]
        lib.trh_set_scattering.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.trh_write_spectrum.argtypes = [C.c_void_p, _abi.c_double_p, C.c_char_p]
        lib.trh_write_spectrum.restype = C.c_int
        lib.trh_write_toomuch.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_int64_p, C.c_char_p]
        lib.trh_write_toomuch.restype = C.c_int
        lib.trh_needs_opacity_build.argtypes = [C.c_void_p]
        lib.trh_needs_opacity_build.restype = C.c_int
        lib.trh_grid_request.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(_abi.c_double_p),
                                         C.POINTER(_abi.c_double_p), C.POINTER(_abi.c_double_p),
                                         C.POINTER(C.c_int32), C.POINTER(_abi.c_int32_p)]
        lib.trh_grid_request.restype = C.c_int
        lib.trh_install_opacity.argtypes = [C.c_void_p, _abi.c_double_p]
        lib.trh_install_opacity.restype = C.c_int
        lib.trh_option.argtypes = [C.c_void_p, C.c_char_p]
        lib.trh_option.restype = C.c_char_p
        lib.trh_messages.argtypes = [C.c_void_p]
        lib.trh_messages.restype = C.c_char_p
        lib.trh_output_plan.argtypes = [C.c_void_p]
        lib.trh_output_plan.restype = C.c_char_p
        lib.trh_option_table.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char)]
        lib.trh_option_table.restype = C.c_int
        lib.trh_wants_detail.argtypes = [C.c_void_p, C.c_int]
        lib.trh_wants_detail.restype = C.c_int
        lib.trh_write_detail.argtypes = [C.c_void_p, C.c_int, _abi.c_double_p]
        lib.trh_write_detail.restype = C.c_int
        lib.trh_write_sample.argtypes = [C.c_void_p, C.c_char_p]
        lib.trh_write_sample.restype = C.c_int
        lib.trh_write_dumps_masked.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p,
                                               _abi.c_int64_p, C.c_char_p]
        lib.trh_write_dumps_masked.restype = C.c_int
        lib.trh_write_ext_dumps.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_double_p, _abi.c_int64_p,
                                            _abi.c_double_p, _abi.c_double_p, _abi.c_double_p, C.c_char_p]
        lib.trh_write_ext_dumps.restype = C.c_int
        _lib = lib
    return _lib


class HostError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s (%d): %s" % (_abi.STATUS.get(code, "?"), code, msg))
        self.code = code


def option_table():
    """[(name, has_arg, kind)] of the host side's option table (same names and order as the
    reference's, argum.c:112-320); kind: 'a' acted on, 'n' no effect as in the reference,
    'w' accepted with a warning, 'x' rejected."""
    lib = _load()
    out, i = [], 0
    while True:
        name, has_arg, kind = C.c_char_p(), C.c_int(), C.c_char()
        if lib.trh_option_table(i, C.byref(name), C.byref(has_arg), C.byref(kind)) != 0:
            return out
        out.append((name.value.decode(), bool(has_arg.value), kind.value.decode()))
        i += 1


class Problem:
    """One parsed transit run: static description + atmosphere + options."""

    def __init__(self, argv: Sequence[str], cwd: Optional[str] = None):
        lib = _load()
        args = [b"transit"] + [a.encode() for a in argv]
        arr = (C.c_char_p * len(args))(*args)
        handle = C.c_void_p()
        err = C.create_string_buffer(512)
        old = os.getcwd()
        try:
            if cwd:
                os.chdir(cwd)      # the reference resolves relative paths against the cwd
            rc = lib.trh_load(len(args), arr, C.byref(handle), err, len(err))
        finally:
            os.chdir(old)
        if rc == 1:
            raise HostError(1, "--help/--version was served (text on stdout); nothing to run")
        if rc != 0:
            raise HostError(rc, err.value.decode(errors="replace"))
        self._h = handle
        self.cwd = cwd or old

    @classmethod
    def from_cfg(cls, cfg: str, extra: Sequence[str] = ()):
        cfg = os.path.abspath(cfg)
        return cls(["-c", os.path.basename(cfg), *extra], cwd=os.path.dirname(cfg))

    def close(self):
        if getattr(self, "_h", None):
            _load().trh_free(self._h)
            self._h = None

    __del__ = close

    # -- plain structs (pointers stay valid while the Problem lives) ---------
    @property
    def static(self) -> _abi.TrxStatic:
        return _load().trh_static(self._h).contents

    @property
    def atm(self) -> _abi.TrxAtm:
        return _load().trh_atm(self._h).contents

    @property
    def opts(self) -> _abi.TrxOpts:
        return _load().trh_opts(self._h).contents

    @property
    def nwn(self) -> int:
        return int(_load().trh_nwn(self._h))

    @property
    def nlayer(self) -> int:
        return int(self.atm.nlayer)

    def wavenumbers(self) -> np.ndarray:
        out = np.empty(self.nwn)
        _load().trh_wavenumbers(self._h, out.ctypes.data_as(_abi.c_double_p))
        return out

    def set_shard(self, lo: int, hi: int):
        _load().trh_set_shard(self._h, lo, hi)

    def option(self, name: str) -> Optional[str]:
        v = _load().trh_option(self._h, name.encode())
        return v.decode() if v is not None else None

    def messages(self):
        """Notes ('I: ...') and warnings ('W: ...') recorded while loading."""
        v = _load().trh_messages(self._h)
        return [m for m in (v.decode(errors="replace") if v else "").split("\n") if m]

    def output_plan(self):
        """{kind: value} of every file a run of this problem writes."""
        v = _load().trh_output_plan(self._h)
        return dict(ln.split(" ", 1) for ln in (v.decode() if v else "").split("\n") if ln)

    def saveext_read(self):
        """--saveext: (e [nlayer][nwn], computed [nlayer]) of the file, or None when there is no valid one
        (restfile_extinct, extinction.c:97-137)."""
        lib = _load()
        lib.trh_saveext_read.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_uint8_p]
        lib.trh_saveext_read.restype = C.c_int
        e = np.zeros((self.nlayer, self.nwn)); c = np.zeros(self.nlayer, dtype=np.uint8)
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = lib.trh_saveext_read(self._h, e.ctypes.data_as(_abi.c_double_p), c.ctypes.data_as(_abi.c_uint8_p))
        finally:
            os.chdir(old)
        if rc < 0:
            raise HostError(rc, "trh_saveext_read")
        return (e, c) if rc == 0 else None

    def saveext_write(self, e: np.ndarray, computed: np.ndarray):
        lib = _load()
        lib.trh_saveext_write.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_uint8_p]
        lib.trh_saveext_write.restype = C.c_int
        e = np.ascontiguousarray(e, dtype=np.float64); c = np.ascontiguousarray(computed, dtype=np.uint8)
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = lib.trh_saveext_write(self._h, e.ctypes.data_as(_abi.c_double_p), c.ctypes.data_as(_abi.c_uint8_p))
        finally:
            os.chdir(old)
        if rc != 0:
            raise HostError(rc, "trh_saveext_write")

    def write_detail(self, which: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = _load().trh_write_detail(self._h, which, a.ctypes.data_as(_abi.c_double_p))
        finally:
            os.chdir(old)
        if rc != 0:
            raise HostError(rc, "write_detail failed")

    def write_sample(self, path: Optional[str] = None):
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = _load().trh_write_sample(self._h, path.encode() if path else None)
        finally:
            os.chdir(old)
        if rc != 0:
            raise HostError(rc, "write_sample failed")

    def write_savefiles(self, out: dict, directory: str):
        """The six `savefiles yes` dumps (tau.c:180-190, 293-335) into `directory`, from a debug
        run's arrays: e, e_cs, tau, last, er, e_scat, e_cloud."""
        d = lambda k: np.ascontiguousarray(out[k], dtype=np.float64).ctypes.data_as(_abi.c_double_p)
        last = np.ascontiguousarray(out["last"], dtype=np.int64)
        lp = last.ctypes.data_as(_abi.c_int64_p)
        rc = _load().trh_write_dumps_masked(self._h, d("e"), d("e_cs"), d("tau"), lp, directory.encode())
        if rc == 0:
            rc = _load().trh_write_ext_dumps(self._h, d("e"), d("e_cs"), lp, d("er"), d("e_scat"), d("e_cloud"),
                                             directory.encode())
        if rc != 0:
            raise HostError(rc, "write_savefiles failed")

    def reload_atm(self, values: np.ndarray):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        rc = _load().trh_reload_atm(self._h, v.ctypes.data_as(_abi.c_double_p), v.size)
        if rc != 0:
            raise HostError(rc, "reload_atm failed")

    def set_radius(self, r):
        _load().trh_set_radius(self._h, float(r))

    def set_cloudtop(self, c):
        _load().trh_set_cloudtop(self._h, float(c))

    def set_scattering(self, flag, logext):
        _load().trh_set_scattering(self._h, int(flag), float(logext))

    def write_spectrum(self, spectrum: np.ndarray, path: Optional[str] = None):
        s = np.ascontiguousarray(spectrum, dtype=np.float64)
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = _load().trh_write_spectrum(self._h, s.ctypes.data_as(_abi.c_double_p),
                                            path.encode() if path else None)
        finally:
            os.chdir(old)
        if rc != 0:
            raise HostError(rc, "write_spectrum failed")

    # -- opacity-grid mode (--opacityfile) -------------------------------------
    @property
    def needs_opacity_build(self) -> bool:
        return bool(_load().trh_needs_opacity_build(self._h))

    def grid_request(self):
        """The (layer x temperature) states calcopacity() sweeps, as the argument
        pack of trx_sweep_permol: (nv, temp, density, zpart, nslot, iso_slot)."""
        nv, nslot = C.c_int32(), C.c_int32()
        t, d, z = _abi.c_double_p(), _abi.c_double_p(), _abi.c_double_p()
        sl = _abi.c_int32_p()
        rc = _load().trh_grid_request(self._h, C.byref(nv), C.byref(t), C.byref(d), C.byref(z),
                                      C.byref(nslot), C.byref(sl))
        if rc != 0:
            raise HostError(rc, "no opacity-grid build is pending")
        return nv.value, t, d, z, nslot.value, sl

    def install_opacity(self, o: np.ndarray):
        """Write the grid file (opacity.c:405-421) and switch to grid mode."""
        o = np.ascontiguousarray(o, dtype=np.float64)
        old = os.getcwd()
        try:
            os.chdir(self.cwd)
            rc = _load().trh_install_opacity(self._h, o.ctypes.data_as(_abi.c_double_p))
        finally:
            os.chdir(old)
        if rc != 0:
            raise HostError(rc, "install_opacity failed")

    # numpy views of the per-layer arrays (copies)
    def layer_arrays(self):
        a = self.atm
        n, nm, ni = a.nlayer, self.static.nmol, self.static.niso
        g = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        return {
            "radius": g(a.radius, n), "temp": g(a.temp, n), "press": g(a.press, n),
            "density": g(a.density, nm * n).reshape(nm, n),
            "zpart": g(a.zpart, ni * n).reshape(ni, n) if ni else np.zeros((0, n)),
        }
