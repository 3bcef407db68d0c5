"""The index arithmetic of csrc/conv_halo.hip EXECUTED on the host: the kernel's address formulas live in
csrc/conv_halo_index.h as host/device functions; tests/conv_halo_host_check.cpp (g++, no HIP, no GPU) drives a lane-level
model of a block with those very functions -- both geometries, the 160-row / tall / K-split variants, strided rows -- and
compares with a direct convolution, exactly.  What tests/test_conv_halo_cpu.py restates in Python, this takes from the
shipped header: a formula mis-typed in the kernel's own source shows up here."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_shipped_index_functions_reproduce_the_convolution(tmp_path):
    gxx = shutil.which("g++") or shutil.which("c++")
    if gxx is None:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "conv_halo_host_check"
    r = subprocess.run([gxx, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "tooncrafter_amd", "csrc"),
                        os.path.join(ROOT, "tests", "conv_halo_host_check.cpp"), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 8 and all(" ok " in l for l in lines), r.stdout


def test_kernel_source_takes_its_formulas_from_the_header():
    """The point of the host check is lost if the kernel keeps private copies of the formulas: it must call the header."""
    src = open(os.path.join(ROOT, "tooncrafter_amd", "csrc", "conv_halo.hip")).read()
    for fn in ("chx::patch_of<", "chx::tiles_m<", "chx::halo_vec<", "chx::frag_a_hp0(", "chx::frag_a_addr(", "chx::frag_b_off(",
               "chx::frag_b_chunk(", "chx::w_lrow(", "chx::w_chunk(", "chx::w_pass_live<", "chx::out_row(", "chx::w_k0(", "chx::tap_shift("):
        assert fn in src, fn
    body = src[src.index("conv_halo_kernel(const TcGemmParams p"):]
    for private in ("pix * 128", "(hp << 7)", "m00 - p.w_out"):
        assert private not in body, f"the kernel still computes `{private}` itself"
