"""ctypes mirror of include/transit_hip.h (plain C structs, no torch types)."""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 5

c_double_p = C.POINTER(C.c_double)
c_int16_p = C.POINTER(C.c_int16)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_float_p = C.POINTER(C.c_float)
c_uint8_p = C.POINTER(C.c_uint8)

SOL_ECLIPSE, SOL_TRANSIT = 0, 1

STATUS = {
    0: "TRX_OK", -1: "TRX_E_ARG", -2: "TRX_E_NOMEM", -3: "TRX_E_HIP", -4: "TRX_E_NODEVICE",
    -5: "TRX_E_RANGE", -6: "TRX_E_UNSUPPORTED", -7: "TRX_E_ORDER", -8: "TRX_E_NOTREACHED",
}


class TrxCia(C.Structure):
    _fields_ = [("nspec", C.c_int32), ("mol", C.c_int32 * 2), ("nwave", C.c_int32),
                ("ntemp", C.c_int32), ("wn", c_double_p), ("temp", c_double_p), ("cs", c_double_p)]


class TrxOpacityGrid(C.Structure):
    _fields_ = [("nmol", C.c_int64), ("ntemp", C.c_int64), ("nlayer", C.c_int64), ("nwave", C.c_int64),
                ("mol_index", c_int32_p), ("temp", c_double_p), ("o", c_double_p)]


class TrxStatic(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("wn_i", C.c_double), ("wn_d", C.c_double), ("nwn", C.c_int64), ("osamp", C.c_int32),
        ("nown", C.c_int64), ("wn_lo", C.c_int64), ("wn_hi", C.c_int64),
        ("ndop", C.c_int32), ("nlor", C.c_int32),
        ("dmin", C.c_float), ("dmax", C.c_float), ("lmin", C.c_float), ("lmax", C.c_float),
        ("timesalpha", C.c_float),
        ("nlines", C.c_int64), ("wl_um", c_double_p), ("isoid", c_int16_p), ("elow", c_double_p),
        ("gf", c_double_p),
        ("niso", C.c_int32), ("iso_mass", c_double_p), ("iso_ratio", c_double_p),
        ("iso_imol", c_int32_p),
        ("nmol", C.c_int32), ("mol_mass", c_double_p), ("mol_radius", c_double_p),
        ("mol_pol", c_double_p), ("mol_is_h2", c_int32_p),
        ("ncia", C.c_int32), ("cia", C.POINTER(TrxCia)),
        ("comm", C.c_void_p), ("nranks", C.c_int32), ("rank", C.c_int32),
        ("ogrid", C.POINTER(TrxOpacityGrid)),
    ]


class TrxAtm(C.Structure):
    _fields_ = [("nlayer", C.c_int32), ("rad_fct", C.c_double), ("radius", c_double_p),
                ("temp", c_double_p), ("press", c_double_p), ("density", c_double_p),
                ("abund", c_double_p), ("zpart", c_double_p)]


class TrxOpts(C.Structure):
    _fields_ = [
        ("solution", C.c_int32), ("toomuch", C.c_double), ("ethresh", C.c_double),
        ("wn_fct", C.c_double), ("nangles", C.c_int32), ("angles_deg", c_double_p),
        ("starrad_cm", C.c_double), ("transparent", C.c_int32), ("modlevel", C.c_int32),
        ("cloud_flag", C.c_int32), ("cloud_ext", C.c_double), ("cloud_top", C.c_double),
        ("cloud_bot", C.c_double), ("cloud_gamma", C.c_double), ("cloud_Q", C.c_double),
        ("cloud_r", C.c_double), ("cloud_sig", C.c_double), ("cloud_refwn", C.c_double),
        ("scat_flag", C.c_int32), ("scat_logext", C.c_double),
        ("layer_chunk", C.c_int32), ("eager", C.c_int32), ("profile", C.c_int32),
    ]


class TrxDebug(C.Structure):
    _fields_ = [("e", c_double_p), ("e_cs", c_double_p), ("tau", c_double_p), ("last", c_int64_p),
                ("intens", c_double_p), ("computed", c_uint8_p),
                ("er", c_double_p), ("e_scat", c_double_p), ("e_cloud", c_double_p)]


class TrxStats(C.Structure):
    _fields_ = [
        ("nlines_inrange", C.c_int64), ("ngroups", C.c_int64), ("nadd", C.c_int64),
        ("layers_swept", C.c_int64), ("neval", C.c_int64), ("nskip", C.c_int64),
        ("sum_bins", C.c_int64), ("table_floats", C.c_int64),
        ("ms_create_table", C.c_double), ("ms_run_total", C.c_double), ("ms_sweep", C.c_double),
        ("ms_k_sweep", C.c_double), ("ms_k_walk", C.c_double), ("ms_k_accum", C.c_double),
        ("sweep_launches", C.c_int64),
        ("ms_tau", C.c_double), ("ms_cia", C.c_double), ("ms_host_total", C.c_double),
        ("ms_spectrum", C.c_double), ("ncandidates", C.c_int64), ("walk_steps", C.c_int64), ("walk_records", C.c_int64), ("walk_record_lanes", C.c_int64),
        ("walk_layers", C.c_int64), ("sum_bins_walk", C.c_int64),
        ("walk_form_steps", C.c_int64 * 3), ("walk_form_layers", C.c_int64 * 3), ("walk_form_record_lanes", C.c_int64 * 3),
        ("walk_form_bins", C.c_int64 * 3), ("ms_k_walk_form", C.c_double * 3), ("ms_walk_span", C.c_double),
    ]

    def as_dict(self):
        return {name: (list(getattr(self, name)) if hasattr(getattr(self, name), "__len__") else getattr(self, name))
                for name, _ in self._fields_}


def bind_engine_api(lib, prefix: str = "trx_"):
    """Declare argtypes/restypes of the create/run/destroy family on *lib*.
    The same struct layouts serve any library exporting this ABI shape."""
    f = lambda n: getattr(lib, prefix + n)
    f("create").argtypes = [C.POINTER(TrxStatic), C.POINTER(C.c_void_p)]
    f("create").restype = C.c_int
    f("run").argtypes = [C.c_void_p, C.POINTER(TrxAtm), C.POINTER(TrxOpts), c_double_p,
                         C.POINTER(TrxDebug)]
    f("run").restype = C.c_int
    f("destroy").argtypes = [C.c_void_p]
    f("destroy").restype = None
    f("get_stats").argtypes = [C.c_void_p, C.POINTER(TrxStats)]
    f("get_stats").restype = C.c_int
    f("table_info").argtypes = [C.c_void_p, c_int64_p, c_int64_p, c_int64_p]
    f("table_info").restype = C.c_int
    f("table_copy").argtypes = [C.c_void_p, c_float_p]
    f("table_copy").restype = C.c_int
    f("width_grids").argtypes = [C.c_void_p, c_double_p, c_double_p]
    f("width_grids").restype = C.c_int
    f("sweep_permol").argtypes = [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_double_p, C.c_double,
                                  C.c_int32, c_int32_p, c_double_p]
    f("sweep_permol").restype = C.c_int
    return lib
