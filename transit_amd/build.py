"""In-tree native builds (explicit compiler calls, no JIT cache).

    libtransit_host.so  g++    host side: options, file formats, samplings
    libtransit_hip.so   hipcc  --offload-arch=gfx950: the kernels + the C ABI
    transit_hip         g++    CLI driver linking both (drop-in for `transit`)

Everything lands in transit_amd/lib/ (git-ignored, shipped to the GPU box with
the tree).  hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or "g++"

HIP_SOURCES = ["hip/trx_api.hip"]
HIP_DEPS = ["hip/trx_kernels.hip.h", "hip/trx_walk.hip.h", "hip/trx_rows.hip.h", "hip/trx_tail.hip.h", "hip/trx_lanes.hip.h", "hip/trx_device.h", "trx_numerics.h", "trx_groups.h"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]


def lib_path(name: str) -> str:
    return os.path.join(LIB, name)


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_host(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(CSRC, "host", "transit_host.cpp")
    out = lib_path("libtransit_host.so")
    deps = [src, os.path.join(CSRC, "trx_numerics.h"), os.path.join(ROOT, "include", "transit_host.h"),
            os.path.join(ROOT, "include", "transit_hip.h")]
    if force or _newer(out, deps):
        _run([CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off",
              "-o", out, src])
    return out


def build_hip(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    out = lib_path("libtransit_hip.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, d) for d in HIP_DEPS] + [os.path.join(ROOT, "include", "transit_hip.h")]
    if force or _newer(out, deps):
        _run([HIPCC, *HIP_FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out, *srcs])
    return out


def build_cli(force: bool = False) -> str:
    out = lib_path("transit_hip")
    src = os.path.join(CSRC, "host", "transit_main.cpp")
    if not os.path.exists(src):
        return ""
    deps = [src, lib_path("libtransit_host.so"), lib_path("libtransit_hip.so")]
    if force or _newer(out, deps):
        _run([CXX, "-O2", "-std=c++17", "-Wall", "-pthread", "-o", out, src, "-I", os.path.join(ROOT, "include"),
              "-L", LIB, "-ltransit_host", "-ltransit_hip", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(force: bool = False):
    build_host(force)
    build_hip(force)
    build_cli(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
