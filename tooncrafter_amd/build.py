"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU).

    python -m tooncrafter_amd.build          # -> tooncrafter_amd/libtooncrafter_hip.so

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtooncrafter_hip.so")
SOURCES = ["gemm.hip", "gemm_wide.hip", "gemm16.hip", "conv_halo.hip", "gemm8.hip", "ff_fused.hip", "tb_fused.hip", "qkv_attn.hip", "gemm_ws.hip", "gemm_mx.hip", "attention.hip", "norm.hip", "elementwise.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(CSRC, "gemm_persist.h"), os.path.join(CSRC, "conv_halo_index.h"),
           os.path.join(ROOT, "include", "tooncrafter_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
if os.environ.get("TC_TIMING_BUILDS") == "1":       # the timing-ablation / interval-trace instantiations (WRONG results by construction:
    FLAGS.append("-DTC_TIMING_BUILDS")              # scripts/*_ablate*, *_trace.py); the product library does not contain them
# per-source flags.  attention.hip: MFMA results straight into VGPRs (gfx950's register file is unified) -- by default
# hipcc parks the score / output accumulators in AGPRs and the in-register softmax then pays 224 v_accvgpr_read/write
# moves per 64-key tile
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _digest() -> str:
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _lib_digest_matches(dig: str) -> bool:
    """The library carries the digest of the sources it was built from (tc_build_info): a stale .so next to
    reverted or newer sources is detected whatever the file times say."""
    try:
        with open(LIB, "rb") as f:
            return ("src:" + dig).encode() in f.read()
    except OSError:
        return False


TORCH_LIB = os.path.join(HERE, "libtooncrafter_torch.so")
TORCH_SRC = os.path.join(CSRC, "torch_ops.cpp")


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """The TORCH_LIBRARY(tooncrafter) operator layer (csrc/torch_ops.cpp): host C++ only, linked against the kernel
    library and libtorch, built in-tree with the host compiler (no hipcc needed: it launches nothing itself)."""
    import torch
    from torch.utils import cpp_extension as ce
    h = hashlib.sha256()
    for p in (TORCH_SRC, os.path.join(ROOT, "include", "tooncrafter_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    dig = h.hexdigest()
    stamp = TORCH_LIB + ".digest"
    if not force and os.path.exists(TORCH_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig \
            and os.path.getmtime(TORCH_LIB) >= os.path.getmtime(LIB):
        return TORCH_LIB
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler for the torch operator layer")
    tlib = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH") or getattr(ce, "ROCM_HOME", None) or "/opt/rocm"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *["-I" + p for p in ce.include_paths()], "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include"),
           TORCH_SRC, "-o", TORCH_LIB, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-L" + HERE, "-ltooncrafter_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return TORCH_LIB


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and _lib_digest_matches(dig):
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), f'-DTC_SRC_DIGEST="{dig}"', "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if not _lib_digest_matches(dig):
        raise RuntimeError("built library does not carry the source digest")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_torch_ops(force="--force" in sys.argv))
