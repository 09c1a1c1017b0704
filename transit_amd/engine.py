"""The spectrum engine: thin ctypes binding of the C ABI in include/transit_hip.h.

`Engine` is the MI355X product path (libtransit_hip.so, hand-written HIP).  It
never falls back to anything else: if the HIP library is missing or no gfx950
device is usable it raises.  `CEngine` is the ABI-shaped binding itself and
serves any library exporting the same create/run/destroy family.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _abi
from .build import lib_path


class EngineError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        super().__init__("%s failed: %s (%d) %s" % (where, _abi.STATUS.get(code, "?"), code, detail))
        self.code = code


class CEngine:
    """create/run/destroy over one handle of an ABI-shaped library."""

    def __init__(self, lib, prefix: str, static: _abi.TrxStatic):
        self._lib, self._p = lib, prefix
        self._h = C.c_void_p()
        self.nwn_total = int(static.nwn)
        self.lo, self.hi = int(static.wn_lo), int(static.wn_hi)
        self.ndop, self.nlor = int(static.ndop), int(static.nlor)
        rc = self._f("create")(C.byref(static), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise EngineError(rc, prefix + "create", self._last_error())

    def _f(self, name):
        return getattr(self._lib, self._p + name)

    def _last_error(self) -> str:
        fn = getattr(self._lib, self._p + "last_error", None)
        if fn is None or not self._h:
            return ""
        fn.argtypes, fn.restype = [C.c_void_p], C.c_char_p
        v = fn(self._h)
        return v.decode(errors="replace") if v else ""

    def close(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    __del__ = close

    @property
    def nwn(self) -> int:
        return self.hi - self.lo if self.hi > self.lo else self.nwn_total

    def run(self, atm: _abi.TrxAtm, opts: _abi.TrxOpts, debug=False,
            n_out: Optional[int] = None) -> Dict[str, np.ndarray]:
        """trx_run.  debug=True also returns every intermediate of trx_debug; a tuple of
        names ("e", "e_cs", "tau", "last", "intens", "computed") returns just those."""
        n = n_out if n_out is not None else self.nwn
        nl, na = int(atm.nlayer), max(int(opts.nangles), 1)
        out = {"spectrum": np.zeros(n)}
        dbg = None
        if debug:
            want = ("e", "e_cs", "tau", "last", "intens", "computed") if debug is True else tuple(debug)
            make = {"e": lambda: np.zeros((nl, n)), "e_cs": lambda: np.zeros((nl, n)),
                    "er": lambda: np.zeros((nl, n)), "e_scat": lambda: np.zeros((nl, n)), "e_cloud": lambda: np.zeros((nl, n)),
                    "tau": lambda: np.zeros((n, nl)), "last": lambda: np.zeros(n, dtype=np.int64),
                    "intens": lambda: np.zeros((na, n)), "computed": lambda: np.zeros(nl, dtype=np.uint8)}
            types = {"last": _abi.c_int64_p, "computed": _abi.c_uint8_p}
            for k in want:
                out[k] = make[k]()
            ptr = lambda k: out[k].ctypes.data_as(types.get(k, _abi.c_double_p)) if k in out else None
            dbg = _abi.TrxDebug(ptr("e"), ptr("e_cs"), ptr("tau"), ptr("last"), ptr("intens"), ptr("computed"),
                                ptr("er"), ptr("e_scat"), ptr("e_cloud"))
        rc = self._f("run")(self._h, C.byref(atm), C.byref(opts),
                            out["spectrum"].ctypes.data_as(_abi.c_double_p),
                            C.byref(dbg) if dbg is not None else None)
        if rc != 0:
            raise EngineError(rc, self._p + "run", self._last_error())
        return out

    def run_into(self, atm: _abi.TrxAtm, opts: _abi.TrxOpts, spectrum: np.ndarray) -> None:
        """trx_run into a caller's float64 array of this handle's shard size: no allocation per call
        (retrieval loops, bench.py's timed step)."""
        if spectrum.dtype != np.float64 or not spectrum.flags.c_contiguous or spectrum.size < self.nwn:
            raise ValueError("spectrum: a C-contiguous float64 array of at least %d elements" % self.nwn)
        rc = self._f("run")(self._h, C.byref(atm), C.byref(opts), spectrum.ctypes.data_as(_abi.c_double_p), None)
        if rc != 0:
            raise EngineError(rc, self._p + "run", self._last_error())

    def sweep_permol(self, nv, temp, density, zpart, ethresh, nslot, iso_slot) -> np.ndarray:
        """computemolext(permol=1) for nv states: returns o[nv][nslot][nwn]."""
        out = np.zeros((nv, nslot, self.nwn))
        rc = self._f("sweep_permol")(self._h, nv, temp, density, zpart, float(ethresh), nslot, iso_slot,
                                     out.ctypes.data_as(_abi.c_double_p))
        if rc != 0:
            raise EngineError(rc, self._p + "sweep_permol", self._last_error())
        return out

    def build_opacity_grid(self, problem) -> np.ndarray:
        """calcopacity(): sweep the (layer x temperature) states of `problem`, let the
        host side write the file and switch the problem to grid mode."""
        nv, t, d, z, nslot, sl = problem.grid_request()
        o = self.sweep_permol(nv, t, d, z, problem.opts.ethresh, nslot, sl)
        problem.install_opacity(o)
        return o

    def restore_extinction(self, e: Optional[np.ndarray], computed: Optional[np.ndarray] = None):
        """trx_restore_extinction: e [nlayer][nwn_shard] and one flag per layer (None: forget them)."""
        fn = self._f("restore_extinction")
        fn.argtypes, fn.restype = [C.c_void_p, C.c_int32, _abi.c_double_p, _abi.c_uint8_p], C.c_int
        if e is None:
            rc = fn(self._h, 0, None, None)
        else:
            e = np.ascontiguousarray(e, dtype=np.float64); c = np.ascontiguousarray(computed, dtype=np.uint8)
            rc = fn(self._h, e.shape[0], e.ctypes.data_as(_abi.c_double_p), c.ctypes.data_as(_abi.c_uint8_p))
        if rc != 0:
            raise EngineError(rc, self._p + "restore_extinction", self._last_error())

    def stats(self) -> Dict[str, float]:
        s = _abi.TrxStats()
        rc = self._f("get_stats")(self._h, C.byref(s))
        if rc != 0:
            raise EngineError(rc, self._p + "get_stats")
        return s.as_dict()

    def table(self):
        """(profsize[ndop,nlor], offset[ndop,nlor], floats[total])"""
        n = self.ndop * self.nlor
        ps, off = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        tot = C.c_int64()
        rc = self._f("table_info")(self._h, ps.ctypes.data_as(_abi.c_int64_p),
                                   off.ctypes.data_as(_abi.c_int64_p), C.byref(tot))
        if rc != 0:
            raise EngineError(rc, self._p + "table_info")
        tab = np.zeros(tot.value, dtype=np.float32)
        rc = self._f("table_copy")(self._h, tab.ctypes.data_as(_abi.c_float_p))
        if rc != 0:
            raise EngineError(rc, self._p + "table_copy")
        return ps.reshape(self.ndop, self.nlor), off.reshape(self.ndop, self.nlor), tab

    def width_grids(self):
        d, l = np.zeros(self.ndop), np.zeros(self.nlor)
        rc = self._f("width_grids")(self._h, d.ctypes.data_as(_abi.c_double_p),
                                    l.ctypes.data_as(_abi.c_double_p))
        if rc != 0:
            raise EngineError(rc, self._p + "width_grids")
        return d, l


_hip = None
HIP_VERSIONS = None


def _one_hip_runtime():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same soname as /opt/rocm's, but loaded by path): if our library came first
    and bound /opt/rocm's copy, a later `import torch` would map a second runtime, and RCCL
    (torch's librccl.so) would then find its HSA uninitialised ("no ROCm-capable device").  So
    when a torch wheel with a bundled runtime is installed, map that copy first -- ours then
    binds to it by soname, whichever import order the caller uses.  torch itself is not imported."""
    import importlib.util
    import sys
    # three busy queues per handle: see INTEGRATION.md section 4 (only effective while the HIP
    # runtime has not initialised yet; multi-GPU launchers should export it themselves)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "torch" in sys.modules or os.environ.get("TRANSIT_AMD_OWN_HIP_RUNTIME"):
        return                                   # torch's runtime is mapped already / the caller wants /opt/rocm's
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def hip_library():
    """Load libtransit_hip.so or raise -- there is no CPU fallback."""
    global _hip
    if _hip is None:
        path = os.environ.get("TRANSIT_HIP_LIB") or lib_path("libtransit_hip.so")   # override: A/B builds
        if not os.path.exists(path):
            raise RuntimeError("HIP extension %s is missing; build it with "
                               "`python -m transit_amd.build` (hipcc --offload-arch=gfx950)" % path)
        _one_hip_runtime()
        lib = C.CDLL(path)
        _abi.bind_engine_api(lib, "trx_")
        lib.trx_run_device.argtypes = [C.c_void_p, C.POINTER(_abi.TrxAtm), C.POINTER(_abi.TrxOpts),
                                       C.c_void_p, C.POINTER(_abi.TrxDebug)]
        lib.trx_run_device.restype = C.c_int
        lib.trx_device_count.restype = C.c_int
        lib.trx_abi_version.restype = C.c_int
        if lib.trx_abi_version() != _abi.ABI_VERSION:
            raise RuntimeError("libtransit_hip.so ABI %d != binding %d" % (lib.trx_abi_version(), _abi.ABI_VERSION))
        # the runtime in this process may be the copy a PyTorch wheel bundles, not the toolchain's the
        # library was compiled against: a different MAJOR version is refused, a different minor one told
        built, running = C.c_int(0), C.c_int(0)
        lib.trx_hip_versions.argtypes, lib.trx_hip_versions.restype = [C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int
        if lib.trx_hip_versions(C.byref(built), C.byref(running)) == 0 and built.value and running.value:
            if built.value // 10000000 != running.value // 10000000:
                raise RuntimeError("libtransit_hip.so was built with HIP %d but runs on runtime %d (set "
                                   "TRANSIT_AMD_OWN_HIP_RUNTIME=1 to keep torch's bundled runtime out)" % (built.value, running.value))
            global HIP_VERSIONS                  # (built, running): 7.2 against the 7.0 of torch's wheel on this image -- same major
            HIP_VERSIONS = (built.value, running.value)
        _hip = lib
    return _hip


class Engine(CEngine):
    """MI355X engine: one handle = one GPU = one wavenumber shard."""

    def __init__(self, static: _abi.TrxStatic):
        super().__init__(hip_library(), "trx_", static)

    def run_device(self, atm, opts, d_spectrum_ptr: int):
        rc = self._lib.trx_run_device(self._h, C.byref(atm), C.byref(opts), C.c_void_p(d_spectrum_ptr), None)
        if rc != 0:
            raise EngineError(rc, "trx_run_device", self._last_error())

    def gather(self, d_slice_ptr: int, d_all_ptr: int, count: int):
        """trx_gather: the one exchange of a sharded job -- every rank's `count` doubles (device
        memory) into d_all in rank order, ncclAllGather over the handle's communicator (a device
        copy without one)."""
        self._lib.trx_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        self._lib.trx_gather.restype = C.c_int
        rc = self._lib.trx_gather(self._h, C.c_void_p(d_slice_ptr), C.c_void_p(d_all_ptr), int(count))
        if rc != 0:
            raise EngineError(rc, "trx_gather", self._last_error())


class Batch:
    """trx_batch: `ways` handles made from one description, each with a host thread of its own inside
    the library; run(atms, opts) deals the atmospheres to them (trx_run_batch) and returns their
    spectra -- each one what Engine.run gives for its atmosphere, bit for bit."""

    def __init__(self, static: _abi.TrxStatic, ways: int = 3):
        lib = hip_library()
        lib.trx_batch_create.argtypes = [C.POINTER(_abi.TrxStatic), C.c_int32, C.POINTER(C.c_void_p)]
        lib.trx_batch_create.restype = C.c_int
        lib.trx_run_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_abi.TrxAtm), C.POINTER(_abi.TrxOpts),
                                      C.POINTER(_abi.c_double_p)]
        lib.trx_run_batch.restype = C.c_int
        lib.trx_batch_destroy.argtypes = [C.c_void_p]
        lib.trx_batch_destroy.restype = None
        lib.trx_last_error.argtypes = [C.c_void_p]
        lib.trx_last_error.restype = C.c_char_p
        self._lib, self._b = lib, C.c_void_p()
        self.nwn = int(static.wn_hi - static.wn_lo)
        rc = lib.trx_batch_create(C.byref(static), int(ways), C.byref(self._b))
        if rc != 0:
            raise EngineError(rc, "trx_batch_create", (lib.trx_last_error(None) or b"").decode(errors="replace"))

    def run(self, atms, opts: _abi.TrxOpts) -> np.ndarray:
        k = len(atms)
        out = np.zeros((k, self.nwn))
        arr = (_abi.TrxAtm * k)(*atms)
        ptrs = (_abi.c_double_p * k)(*[out[j].ctypes.data_as(_abi.c_double_p) for j in range(k)])
        rc = self._lib.trx_run_batch(self._b, k, arr, C.byref(opts), ptrs)
        if rc != 0:
            raise EngineError(rc, "trx_run_batch", (self._lib.trx_last_error(None) or b"").decode(errors="replace"))
        return out

    def close(self):
        if self._b:
            self._lib.trx_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


LOG_FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
_log_keep = None


def set_log(callback, max_level: int = 3):
    """Route the library's messages (trx_set_log) to callback(level, text); None = silent."""
    global _log_keep
    lib = hip_library()
    lib.trx_set_log.argtypes = [LOG_FN, C.c_void_p, C.c_int]
    lib.trx_set_log.restype = None
    if callback is None:
        _log_keep = LOG_FN(0)
    else:
        _log_keep = LOG_FN(lambda lvl, msg, _u: callback(int(lvl), msg.decode(errors="replace")))
    lib.trx_set_log(_log_keep, None, int(max_level))


def device_count() -> int:
    return int(hip_library().trx_device_count())
