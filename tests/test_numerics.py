"""trx_numerics.h building blocks that replace an operation of the reference by a cheaper
sequence must return that operation's bits (CPU: the header is shared by host and kernels)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quotient_rn_is_the_division(tmp_path):
    """x * (1/d) with two fused residual corrections == x / d, bit for bit: divisors 6, layer
    spacings and 2 step^2 in cm, mantissas next to all-ones; numerators random, exact multiples
    and their neighbours, zero (eclipse.c:66-80 through k_optical_depth_vertical)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "quotient_check")
    try:
        hw_fma = " fma " in open("/proc/cpuinfo").read()
    except OSError:
        hw_fma = False
    subprocess.run([gxx, "-O2", "-ffp-contract=off"] + (["-mfma"] if hw_fma else []) + ["-I", os.path.join(ROOT, "transit_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "quotient_check.cpp")], check=True)
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "20000000 operands, 0 differ" in out.stdout
    assert "1000000 parabolas, 0 differ" in out.stdout      # parab3 == parab3_recip == parab3_chain, bit for bit


def test_threaded_grouping_is_the_sequential_loop(tmp_path):
    """trx_create's co-added groups, cut into pieces and grouped on several threads, and its
    per-bin counts == the one-thread loops of extinction.c:445-462 (tests/groups_check.cpp:
    sparse, grid-dense and over-dense lists, many small isotope blocks, ties, lines out of range)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "groups_check")
    subprocess.run([gxx, "-O2", "-pthread", "-I", os.path.join(ROOT, "transit_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "groups_check.cpp")], check=True)
    out = subprocess.run([exe, "40"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert " 0 differ" in out.stdout
