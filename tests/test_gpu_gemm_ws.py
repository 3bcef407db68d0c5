"""Weight-stationary K = 320 GEMM (csrc/gemm_ws.hip; reference call sites: the level-0 nn.Linear layers of
lvdm/modules/attention.py:53-57,75-76,242-246,269,290,418-438 and the LayerNorms in front of them, :225-227) against the
fp32 statement of the operator (tests/emu_ops.py) and against the tiled kernel it replaces.

Cases: every instantiation (plain / residual / GEGLU, each with and without the LayerNorm prologue), row counts with a
ragged last tile, fewer tiles than persistent blocks, more tiles than blocks, strided A / C / residual views, and the
debug wait mode (TC_GEMM_WS=3: vmcnt(0) everywhere) which must give the same bits as the counted waits.
"""
import os

import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"
K = 320


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


class ws_mode:
    """TC_GEMM_WS is read per call by the library."""

    def __init__(self, v):
        self.v = str(v)

    def __enter__(self):
        self.old = os.environ.get("TC_GEMM_WS")
        os.environ["TC_GEMM_WS"] = self.v

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("TC_GEMM_WS", None)
        else:
            os.environ["TC_GEMM_WS"] = self.old


def _case(m, n, geglu, seed):
    a = rnd(m, K, seed=seed, scale=1.0) + 0.25                 # non-zero mean: the LayerNorm prologue has work to do
    a = a.to(BF16)
    if geglu:
        from tooncrafter_amd.lvdm.common import pack_geglu
        w32 = rnd(n, K, seed=seed + 1, scale=K ** -0.5, dtype=torch.float32)
        b32 = rnd(n, seed=seed + 2, dtype=torch.float32)
        w, bias = pack_geglu(w32, b32)
    else:
        w = rnd(n, K, seed=seed + 1, scale=K ** -0.5)
        bias = rnd(n, seed=seed + 2, dtype=torch.float32)
    return a, w, bias


@pytest.mark.parametrize("m", [64, 130, 8192, 8250, 40960])
@pytest.mark.parametrize("n,geglu,res", [(320, False, False), (320, False, True), (960, False, False), (2560, True, False),
                                          (512, True, False)])
@pytest.mark.parametrize("ln", [False, True])
def test_gemm_ws_vs_spec(hip, emu, m, n, geglu, res, ln):
    a, w, bias = _case(m, n, geglu, seed=100 + m % 97)
    n_out = n // 2 if geglu else n
    residual = rnd(m, n_out, seed=5) if res else None
    kw = dict(act=ACT_GEGLU if geglu else ACT_NONE, residual=residual)
    if ln:
        kw["a_norm_eps"] = 1e-5
    with ws_mode(2):                                           # 2 = whenever the shape allows (also below 8192 rows)
        assert hip.gemm_ln_eligible(m, n, K, geglu=geglu)
        out = hip.gemm(a, w, bias, **kw)
    check(out, emu.gemm(a, w, bias, **kw), f"gemm_ws m={m} n={n} geglu={geglu} res={res} ln={ln}")
    if not ln:
        with ws_mode(0):                                       # the tiled kernels on the same problem
            old = hip.gemm(a, w, bias, **kw)
        check(out, old, f"gemm_ws vs tiled kernel m={m} n={n}", rel=3e-3)
    with ws_mode(3):                                           # vmcnt(0) waits: same arithmetic, same bits
        safe = hip.gemm(a, w, bias, **kw)
    assert torch.equal(out, safe), "counted-vmcnt and vmcnt(0) runs differ: a tile was read before it landed"
    with ws_mode(4):                                           # deeper store window (the default count of the heuristic)
        deep = hip.gemm(a, w, bias, **kw)
    assert torch.equal(out, deep), "the deeper store window changed the result: a tile was read before it landed"


def test_gemm_ws_strided_views_and_bounds(hip, emu):
    """A, C and the residual as column slices of wider buffers; the columns next to C must stay untouched, and so must
    the rows behind M (the last tile is ragged)."""
    m, n = 8250, 320
    big_a = rnd(m, 3 * K, seed=21)
    a = big_a[:, K:2 * K]
    w, bias = rnd(n, K, seed=22, scale=K ** -0.5), rnd(n, seed=23, dtype=torch.float32)
    big_r = rnd(m, 2 * n, seed=24)
    res = big_r[:, n:]
    outbuf = torch.full((m + 70, 3 * n), 7.0, dtype=BF16, device=DEV)
    with ws_mode(2):
        hip.gemm(a, w, bias, residual=res, out=outbuf[:m, n:2 * n])
    check(outbuf[:m, n:2 * n], emu.gemm(a, w, bias, residual=res), "gemm_ws strided A / C / residual")
    assert float((outbuf[:m, :n] - 7).abs().max()) == 0 and float((outbuf[:m, 2 * n:] - 7).abs().max()) == 0
    assert float((outbuf[m:] - 7).abs().max()) == 0, "rows behind M were written"


def test_gemm_ws_transpose_detecting(hip):
    """Identity-like A against an asymmetric W: a swapped fragment / C-write mapping cannot pass."""
    m, n = 320, 320
    a = torch.eye(m, K, device=DEV, dtype=BF16)
    w = ((torch.arange(n, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] % 7).float())
    w = (w / w.max()).to(BF16)
    with ws_mode(2):
        out = hip.gemm(a, w)
    assert torch.equal(out, w.t().contiguous()), "fragment or C-write layout is wrong"


def test_gemm_ws_layernorm_rows_with_large_mean(hip):
    """The prologue's two-pass variance against float64 on rows with mean >> std (|mean| = 60, std = 0.5)."""
    m, n = 8192, 320
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(m, K, generator=g) * 0.5 + 60.0).to(BF16).to(DEV)
    w = rnd(n, K, seed=31, scale=K ** -0.5)
    with ws_mode(2):
        out = hip.gemm(x, w, a_norm_eps=1e-5)
    xd = x.double()
    xn = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)
    ref = xn.to(BF16).double() @ w.double().t()
    err = float((out.double() - ref).norm() / ref.norm())
    print(f"gemm_ws LayerNorm prologue, mean 60 / std 0.5: rel-L2 {err:.3e}")
    assert err < 6e-3


def test_a_norm_refused_elsewhere(hip):
    a, w = rnd(256, 640, seed=1), rnd(640, 640, seed=2)
    assert not hip.gemm_ln_eligible(256, 640, 640)
    with pytest.raises(ValueError):
        hip.gemm(a, w, a_norm_eps=1e-5)
