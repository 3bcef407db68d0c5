"""The C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every symbol include/tooncrafter_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

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
                         ("TcDdimParams", _lib.TcDdimParams), ("TcGemmMxParams", _lib.TcGemmMxParams),
                         ("TcFfParams", _lib.TcFfParams), ("TcTbParams", _lib.TcTbParams), ("TcTqaParams", _lib.TcTqaParams)):
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


def test_fp8_routing_rule(monkeypatch):
    """TC_FP8 routing (BASELINE configs[4]): which GEMM launches take the MXFP8 kernel -- host logic only."""
    from tooncrafter_amd._lib import GATHER_CONV3x3, GATHER_LINEAR, TcGemmParams
    from tooncrafter_amd.ops import HipOps
    assert HipOps().fp8 is None                              # off unless asked for
    monkeypatch.setenv("TC_FP8", "1")
    assert HipOps().fp8 == "linear"                          # the default fp8 mode: wide-N projections only
    monkeypatch.setenv("TC_FP8", "all")
    h = HipOps()
    assert h.fp8 == "all"

    def prm(m, n, k, gather=GATHER_LINEAR, cin=0):
        p = TcGemmParams()
        p.m, p.n, p.k, p.gather, p.cin = m, n, k, gather, cin
        return p
    assert h._fp8_eligible(prm(20480, 1920, 640), False, 1920, 1)            # level-1 qkv: wide N
    assert not h._fp8_eligible(prm(81920, 320, 1280), False, 320, 1)         # ff2: the quantiser costs more than it saves
    assert not h._fp8_eligible(prm(5120, 1280, 5120), False, 1280, 1)        # level-2 ff2 / out-projection: N < 2 K
    assert not h._fp8_eligible(prm(5120, 1280, 1280), False, 1280, 1)
    assert not h._fp8_eligible(prm(81920, 960, 320), False, 960, 1)          # short K
    assert not h._fp8_eligible(prm(20480, 1920, 640), False, 1920, 2)        # batched
    assert not h._fp8_eligible(prm(77, 2048, 1024), False, 2048, 1)          # a handful of rows (context projections)
    assert not h._fp8_eligible(prm(20480, 1284, 640), False, 1284, 1)        # N tail: the scalar epilogue stays bf16
    assert h._fp8_eligible(prm(81920, 320, 2880, GATHER_CONV3x3, 320), True, 320, 1)
    assert not h._fp8_eligible(prm(20480, 640, 17280, GATHER_CONV3x3, 1920), True, 640, 1)   # cin > 1280
    monkeypatch.setenv("TC_FP8", "conv")
    h = HipOps()
    assert not h._fp8_eligible(prm(20480, 1920, 640), False, 1920, 1)
    assert h._fp8_eligible(prm(81920, 320, 2880, GATHER_CONV3x3, 320), True, 320, 1)


def test_prefetch_rule_is_one_host_function():
    """ABI 12: which consumer weights a norm launch streams ahead is decided by ops.HipOps.prefetch_list alone (both
    bindings call it): small-M consumers only, tensors worth a request, at most TC_PREFETCH_MAX / 96 MB per launch."""
    import torch
    from tooncrafter_amd import _lib
    from tooncrafter_amd.ops import HipOps
    h = HipOps()
    big = [torch.empty(1280, 11520, dtype=torch.bfloat16), torch.empty(1280, 1280, dtype=torch.bfloat16)]      # 28 MB, 3.1 MB
    small = torch.empty(320, 320, dtype=torch.bfloat16)                                                      # 0.2 MB
    assert [t.data_ptr() for t in h.prefetch_list(5120, big)] == [t.data_ptr() for t in big]
    assert h.prefetch_list(5120, [small, None, big[1]]) == [big[1]]                   # too small / absent: skipped
    assert h.prefetch_list(20480, big) == [] and h.prefetch_list(81920, big) == []  # levels 0 / 1: W is small beside A
    assert h.prefetch_list(5120, None) == [] and h.prefetch_list(5120, []) == []
    assert len(h.prefetch_list(1280, [big[1]] * 7)) == _lib.TC_PREFETCH_MAX
    assert h.prefetch_list(1280, [big[0]] * 4) == [big[0]] * 3                        # 96 MB per launch
    assert h.prefetch_list(1280, [big[0].t()]) == []                                  # non-contiguous views are not streamed
    assert h.prefetch_list(1280, [big[0].view(-1)[1:1 + (1 << 20)]]) == []            # nor is a buffer that is not 16-byte aligned
    h.prefetch_on = False
    assert h.prefetch_list(5120, big) == []
    with pytest.raises(_lib.TooncrafterHipError):
        HipOps._prefetch_struct(big)                                                  # CPU tensors never cross the ABI
