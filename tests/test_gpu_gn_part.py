"""GroupNorm statistics from the producing GEMM (ABI 9: TcGemmParams.gn_part / tc_gemm_gn_rows / tc_groupnorm_part;
reference lvdm/basics.py:76-87 over the outputs of lvdm/modules/networks/openaimodel3d.py:154,179,255-266).

The GEMM epilogues of the 160x160-tile kernel (csrc/gemm16.hip) and of the 128x128-tile kernel (csrc/gemm.hip) emit, per
block of 160 / 128 output rows and per column, the sum and the sum of squares of the bf16-ROUNDED values they store;
tc_groupnorm_part reduces them (fp64) instead of reading the tensor once more.  Checked here: the partial sums against
float64 sums of the stored tensor, the normalised result against the two-pass operator and against the fp32 statement,
rows with mean >> std, ragged shapes, and that problems whose kernel cannot emit statistics say so (None).
"""
import os

import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    """The operator set with producer statistics ON (TC_GN_PART=1; the product default is off: profiles/r04_gn_part_ab.txt)."""
    from tooncrafter_amd.ops import HipOps
    h = HipOps()
    h.gn_part = True
    return h


@pytest.fixture(autouse=True)
def _implicit_gemm_convolutions(monkeypatch):
    """Producer statistics come from the implicit-GEMM kernels' epilogues; the tap-reuse kernel that takes the UNet's 3x3
    convolutions by default (csrc/conv_halo.hip) emits none (tc_gemm_gn_rows says 0 there).  These tests are about the
    emitting kernels, so the 3x3 convolutions are kept on them."""
    monkeypatch.setenv("TC_CONV_HALO", "0")


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def _c3(frames, h, w, cin):
    return dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)


def _ct(frames, hw, cin):
    return dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=1, w_out=hw)


def _check_part(out, part, tag):
    """part.sums[blk, 0 / 1, col] == sum / sum of squares of out[blk * rows : (blk + 1) * rows, col] (float64 truth)."""
    m, n = out.shape
    r = part.rows
    nb = (m + r - 1) // r
    assert tuple(part.sums.shape) == (nb, 2, n) and part.of is out
    x = torch.zeros((nb * r, n), dtype=torch.float64, device=out.device)
    x[:m] = out.double()
    x = x.reshape(nb, r, n)
    s, q = x.sum(1), (x * x).sum(1)
    es = float((part.sums[:, 0].double() - s).abs().max() / (s.abs().max() + 1e-30))
    eq = float((part.sums[:, 1].double() - q).abs().max() / (q.abs().max() + 1e-30))
    print(f"{tag}: block rows {r}, {nb} blocks; partial sums vs float64: sum {es:.2e}, sum of squares {eq:.2e}")
    assert es < 2e-5 and eq < 2e-5, (tag, es, eq)


CASES = [
    # tag, m, n, cin, conv, residual, row_bias, expected block height (None: no statistics from this kernel)
    ("3x3 level-0 (160-tile kernel)", 32 * 40 * 64, 320, 320, _c3(32, 40, 64, 320), True, True, 160),
    ("t3 level-0 (160-tile kernel)", 32 * 2560, 320, 320, _ct(32, 2560, 320), False, False, 160),
    ("3x3 level-1 (160-tile kernel)", 32 * 20 * 32, 640, 640, _c3(32, 20, 32, 640), True, False, 160),
    ("t3 level-2 (128-tile kernel)", 32 * 160, 1280, 1280, _ct(32, 160, 1280), False, False, 128),
    ("linear 16384 x 512 x 512 (128-tile kernel)", 16384, 512, 512, None, True, False, 128),
    ("ragged 3x3 5 x 17 x 23, 128 -> 264 (128-tile kernel, N and M tails)", 5 * 17 * 23, 264, 128, _c3(5, 17, 23, 128), True, True, None),
]


@pytest.mark.parametrize("tag,m,n,cin,conv,res,rb,want_rows", CASES, ids=[c[0] for c in CASES])
def test_partial_sums_and_norm(hip, emu, tag, m, n, cin, conv, res, rb, want_rows):
    taps = 1 if conv is None else (9 if conv["kind"] == "3x3" else 3)
    k = cin * taps
    a = rnd(m, cin, seed=3) + 0.25
    a = a.to(BF16)
    w, bias = rnd(n, k, seed=4, scale=k ** -0.5), rnd(n, seed=5, dtype=torch.float32)
    residual = rnd(m, n, seed=6) if res else None
    frames = conv["frames"] if conv else 8
    row_div = m // frames
    row_bias = rnd(frames, n, seed=7, dtype=torch.float32) if rb else None
    kw = dict(conv=conv, residual=residual, row_bias=row_bias, row_div=row_div if rb else 0)
    out, part = hip.gemm(a, w, bias, gn_stats=True, **kw)
    plain = hip.gemm(a, w, bias, **kw)
    assert torch.equal(out, plain), "asking for statistics changed the result"
    if want_rows is None and part is None:
        return                                                   # a small problem on the 64x64 tiles: no statistics, said so
    assert part is not None and (want_rows is None or part.rows == want_rows), (tag, None if part is None else part.rows)
    _check_part(out, part, tag)
    # the norm from the partial sums == the two-pass norm (statistics agree to fp32 rounding: at most 1 bf16 ulp apart)
    gamma, beta = rnd(n, seed=8, dtype=torch.float32) + 1.0, rnd(n, seed=9, dtype=torch.float32)
    for samples in (frames, max(1, frames // 16)):
        rows = m // samples
        if rows % part.rows or n % 32:
            continue
        y1 = hip.groupnorm(out, gamma, beta, samples=samples, rows=rows, eps=1e-5, silu=True, part=part)
        y0 = hip.groupnorm(out, gamma, beta, samples=samples, rows=rows, eps=1e-5, silu=True)
        check(y1, emu.groupnorm(out, gamma, beta, samples=samples, rows=rows, eps=1e-5, silu=True), f"{tag}: norm from partial sums, {samples} samples")
        d = float((y1.float() - y0.float()).abs().max())
        print(f"{tag}: {samples} samples x {rows} rows: max |part - two-pass| {d:.3e}")
        assert d <= 2.0 ** -6 * float(y0.float().abs().max()), d


def test_rows_with_large_mean(hip):
    """mean 40, std 0.5 per channel: E[x^2] - mean^2 from fp32 block sums, reduced in fp64, against float64 GroupNorm."""
    frames, h, w, c = 32, 40, 64, 320
    m = frames * h * w
    conv = _c3(frames, h, w, c)
    a = (rnd(m, c, seed=11) * 0.05).to(BF16)
    wgt = rnd(c, 9 * c, seed=12, scale=0.02)
    bias = torch.full((c,), 40.0, device=DEV)
    out, part = hip.gemm(a, wgt, bias, conv=conv, gn_stats=True)
    assert part is not None
    gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    y = hip.groupnorm(out, gamma, beta, samples=frames, rows=h * w, eps=1e-5, part=part)
    x = out.double().reshape(frames, h * w, 32, c // 32)
    ref = (x - x.mean(dim=(1, 3), keepdim=True)) / torch.sqrt(x.var(dim=(1, 3), unbiased=False, keepdim=True) + 1e-5)
    err = float((y.double() - ref.reshape(m, c)).norm() / ref.norm())
    print(f"GroupNorm from partial sums, mean 40 / std {float(x.std()):.3f}: rel-L2 {err:.3e}")
    assert err < 6e-3


def test_switch_and_unsupported_routes(hip):
    a, w = rnd(1280, 1280, seed=21), rnd(1280, 1280, seed=22, scale=0.03)
    out, part = hip.gemm(a, w, gn_stats=True)                   # 100 tiles of 128 x 128: the 64x64-tile family -> none
    assert part is None and out.shape == (1280, 1280)
    old = os.environ.get("TC_GN_PART")
    os.environ["TC_GN_PART"] = "0"
    try:
        out, part = hip.gemm(rnd(4096, 512, seed=23), rnd(512, 512, seed=24, scale=0.05), gn_stats=True)
        assert part is None
    finally:
        if old is None:
            os.environ.pop("TC_GN_PART", None)
        else:
            os.environ["TC_GN_PART"] = old
