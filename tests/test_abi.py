"""The C-ABI libraries load and export every symbol their headers declare
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from transit_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, txt)))


def test_host_library_exports():
    lib = ctypes.CDLL(build.build_host())
    names = declared("transit_host.h", "trh_")
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_hip_library_exports():
    path = build.lib_path("libtransit_hip.so")
    if not os.path.exists(path):
        build.build_hip()
    lib = ctypes.CDLL(path)
    names = declared("transit_hip.h", "trx_")
    assert {"trx_create", "trx_run", "trx_destroy", "trx_strerror"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n
    lib.trx_abi_version.restype = ctypes.c_int
    assert lib.trx_abi_version() == 5
    lib.trx_strerror.restype = ctypes.c_char_p
    lib.trx_strerror.argtypes = [ctypes.c_int]
    assert b"sorted" in lib.trx_strerror(-7)


def test_hip_code_object_is_gfx950():
    path = build.lib_path("libtransit_hip.so")
    if not os.path.exists(path):
        build.build_hip()
    blob = open(path, "rb").read()
    assert b"gfx950" in blob
    for k in (b"k_line_walk", b"k_walk_combine", b"k_layer_max", b"k_group_sweep", b"k_sticky_index", b"k_accumulate",
              b"k_optical_depth", b"k_emission", b"k_modulation", b"k_voigt_bins"):
        assert k in blob, k


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "transit_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "trx_oracle" not in txt and "libtrx_oracle" not in txt, os.path.join(dp, f)
                assert "oracle_lib" not in txt, os.path.join(dp, f)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cases import golden
    from transit_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(golden("eclipse_small").problem.static)
