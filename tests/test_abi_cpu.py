"""The C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every symbol include/tooncrafter_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from tooncrafter_amd import _lib, build
    path = build.build(force=False, verbose=False)
    assert os.path.exists(path)
    with open(os.path.join(ROOT, "include", "tooncrafter_hip.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in the header"
    lib = ctypes.CDLL(path)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"library lacks {missing}"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    loaded = _lib.load()
    assert loaded.tc_abi_version() == _lib.TC_ABI_VERSION
    assert b"gfx950" in loaded.tc_build_info()


def test_struct_layout_matches_header():
    """ctypes mirrors of the parameter structs have the field order of the header."""
    from tooncrafter_amd import _lib
    with open(os.path.join(ROOT, "include", "tooncrafter_hip.h")) as f:
        header = f.read()
    for cname, ctype in (("TcGemmParams", _lib.TcGemmParams), ("TcAttnParams", _lib.TcAttnParams),
                         ("TcDdimParams", _lib.TcDdimParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
        assert names == [f[0] for f in ctype._fields_], (cname, names)


def test_ops_refuse_cpu_tensors():
    """The product path fails loudly without the GPU: no silent CPU fallback."""
    import pytest
    import torch
    from tooncrafter_amd._lib import TooncrafterHipError
    from tooncrafter_amd.ops import HipOps
    hip = HipOps()
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
