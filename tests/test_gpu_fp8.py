"""MXFP8 GEMM path (BASELINE.json configs[4]) on the MI355X, through the C ABI.

Two statements are tested separately:
 1. tc_quant_mxfp8 is the OCP MX quantiser of oracle/mx.py BIT FOR BIT (e4m3 bytes and E8M0 scale bytes);
 2. tc_gemm_mxfp8 computes, for bf16 inputs (a, w), exactly epilogue(gather(fq(a)) @ fq(w)^T) where fq() is the
    round trip through MXFP8 -- fq values are exactly representable in bf16, so the plain-PyTorch emulation of
    the bf16 operator (tests/emu_ops.py) applied to fq(a), fq(w) is the reference, at the bf16-path tolerances
    (rel-L2 <= 4e-3 with bf16 outputs, <= 1e-4 with fp32 outputs: only the fp32 summation order differs).
The error the FORMAT introduces (fq(x) vs x) is not a kernel property; it is measured end to end on the
full-size UNet in test_unet_full_size_fp8 below and bounded there."""
import os

import numpy as np
import pytest
import torch

from emu_ops import EmuOps
from oracle import mx
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU
from test_gpu_ops import check, rnd

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


@pytest.fixture(scope="module")
def hip8():
    from tooncrafter_amd.ops import HipOps
    h = HipOps()
    h.fp8, h.fp8_min_k, h.fp8_min_m, h.fp8_min_n, h.fp8_max_cin, h.fp8_n_over_k = "all", 0, 1, 0, 1 << 20, 0.0
    return h


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def fq(x):
    """bf16 tensor -> its MXFP8 round trip along the last dimension, as bf16 (exact)."""
    y = mx.fake_quant(x.detach().cpu().float())
    assert torch.equal(y.to(BF16).float(), y)
    return y.to(BF16).to(x.device)


def ran_mx(h, before):
    assert h.fp8_calls["mx"] == before["mx"] + 1 and h.fp8_calls["bf16"] == before["bf16"], "the fp8 kernel did not run"


@pytest.mark.parametrize("rows,k,ld", [(257, 320, 320), (64, 32, 64), (1000, 1280, 1280), (33, 2560, 2688), (5, 96, 96)])
def test_quantiser_is_the_mx_restatement_bit_for_bit(hip8, rows, k, ld):
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, ld, generator=g) * torch.exp(4.0 * torch.randn(rows, 1, generator=g))
    x[:, :ld // 2] *= torch.exp(2.0 * torch.randn(1, ld // 2, generator=g))        # spread inside the blocks too
    x[0, :32] = 0.0                                                                # an all-zero block
    x[1 % rows, 5] = 3.0e38                                                        # bf16 near its maximum
    x[2 % rows, 40] = 1.0e-38                                                      # denormal neighbourhood
    xb = x.to(BF16).to(DEV)
    q, s = hip8.quant_mxfp8(xb[:, :k] if ld != k else xb, k)
    qr, sr = mx.quantize_mxfp8(xb[:, :k].cpu().float())
    nblk = k // 32
    assert s.shape == (rows, (k + 127) // 128 * 4)
    assert torch.equal(s[:, :nblk].cpu(), sr), "E8M0 scale bytes differ"
    assert int(s[:, nblk:].cpu().sum()) == 0, "scale padding columns must be zero"
    qh = q.cpu()
    same = qh == qr
    # signed zero is one value: allow 0x80 vs 0x00
    zero = ((qh & 0x7f) == 0) & ((qr & 0x7f) == 0)
    assert bool((same | zero).all()), f"{int((~(same | zero)).sum())} e4m3 bytes differ"


@pytest.mark.parametrize("m,n,k,f32", [(1024, 128, 128, False), (1500, 320, 320, False), (1100, 136, 96, True),
                                        (4096, 1280, 2560, False), (2048, 640, 1280, True), (70, 64, 64, True)])
def test_gemm_mx_linear(hip8, emu, m, n, k, f32):
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5)
    bias = rnd(n, seed=3, dtype=torch.float32)
    c0 = dict(hip8.fp8_calls)
    out = hip8.gemm(a, w, bias, out_f32=f32)
    ran_mx(hip8, c0)
    check(out, emu.gemm(fq(a), fq(w), bias, out_f32=f32), f"mx linear {m}x{n}x{k}", f32=f32)


def test_gemm_mx_transpose_and_block_scale_detecting(hip8, emu):
    """Asymmetric operands whose magnitude varies per row AND per K block: a swapped scale byte, a K block applied
    to the wrong half of the lanes or a transposed output tile all change the result by orders of magnitude."""
    m, n, k = 1280, 384, 640
    g = torch.Generator().manual_seed(5)
    a = torch.randn(m, k, generator=g) * torch.exp2(torch.randint(-6, 7, (m, k // 32), generator=g).float()).repeat_interleave(32, 1)
    w = torch.randn(n, k, generator=g) * torch.exp2(torch.randint(-6, 7, (n, k // 32), generator=g).float()).repeat_interleave(32, 1)
    a, w = a.to(BF16).to(DEV), (w * k ** -0.5).to(BF16).to(DEV)
    out = hip8.gemm(a, w, None, out_f32=True)
    check(out, emu.gemm(fq(a), fq(w), None, out_f32=True), "mx linear, per-block magnitudes", f32=True)


def test_gemm_mx_epilogues(hip8, emu):
    m, n, k, hw = 2048, 320, 640, 256
    a, w = rnd(m, k, seed=6), rnd(n, k, seed=7, scale=k ** -0.5)
    bias = rnd(n, seed=8, dtype=torch.float32)
    rb = rnd(m // hw, n, seed=9, dtype=torch.float32)
    res = rnd(m, n, seed=10)
    kw = dict(act=ACT_SILU, row_bias=rb, row_div=hw, residual=res, alpha=0.7, out_scale=1.3)
    check(hip8.gemm(a, w, bias, **kw), emu.gemm(fq(a), fq(w), bias, **kw), "mx epilogue: silu/row-bias/residual/scales")
    wide = torch.zeros((m, 2 * k), dtype=BF16, device=DEV)
    wide[:, k:] = a                                                                 # column-sliced A (lda > k)
    check(hip8.gemm(wide[:, k:], w, bias, residual=res), emu.gemm(fq(a), fq(w), bias, residual=res), "mx strided A")


def test_gemm_mx_geglu(hip8):
    from tooncrafter_amd.lvdm.common import pack_geglu
    m, c = 1536, 640
    x = rnd(m, c, seed=11)
    w = rnd(8 * c, c, seed=12, scale=c ** -0.5, dtype=torch.float32)
    b = rnd(8 * c, seed=13, dtype=torch.float32)
    wp, bp = pack_geglu(w, b)
    c0 = dict(hip8.fp8_calls)
    out = hip8.gemm(x, wp, bp, act=ACT_GEGLU)
    ran_mx(hip8, c0)
    full = fq(x).float() @ fq(w.to(BF16)).float().t() + b
    v, gate = full.chunk(2, dim=-1)
    check(out, (v * torch.nn.functional.gelu(gate)).to(BF16), "mx GEGLU (packed weights) vs unpacked definition")


@pytest.mark.parametrize("frames,h,w,cin,cout,stride,ups", [
    (2, 8, 8, 64, 64, 1, False), (3, 5, 8, 128, 320, 1, False), (2, 10, 16, 64, 128, 2, False),
    (2, 7, 9, 64, 64, 2, False), (2, 5, 8, 128, 128, 1, True), (1, 40, 64, 320, 320, 1, False),
    (2, 12, 10, 960, 320, 1, False), (1, 9, 11, 320, 640, 2, False)])
def test_gemm_mx_conv3x3(hip8, emu, frames, h, w, cin, cout, stride, ups):
    """cin = 320 / 960: a 128-element K-step straddles two taps (per-lane tap decode)."""
    x = rnd(frames * h * w, cin, seed=14)
    wt = rnd(cout, 9 * cin, seed=15, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=16, dtype=torch.float32)
    ho = h * 2 if ups else (h - 1) // stride + 1
    wo = w * 2 if ups else (w - 1) // stride + 1
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, upsample=ups)
    c0 = dict(hip8.fp8_calls)
    o = hip8.gemm(x, wt, bias, conv=geom)
    ran_mx(hip8, c0)
    check(o, emu.gemm(fq(x), fq(wt), bias, conv=geom), f"mx conv3x3 f{frames} {h}x{w} {cin}->{cout} s{stride} ups{ups}")


def test_gemm_mx_conv3x3_asymmetric_pad(hip8, emu):
    frames, h, w, cin, cout = 2, 10, 12, 128, 128
    x = rnd(frames * h * w, cin, seed=30)
    wt = rnd(cout, 9 * cin, seed=31, scale=(9 * cin) ** -0.5)
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h // 2, w_out=w // 2, stride=2, upsample=False, pad=0)
    check(hip8.gemm(x, wt, None, conv=geom), emu.gemm(fq(x), fq(wt), None, conv=geom), "mx conv3x3 stride 2, pad (0,1,0,1)")


@pytest.mark.parametrize("b,t,hw,c", [(1, 16, 40, 64), (2, 4, 64, 128), (1, 3, 24, 320), (2, 1, 16, 64)])
def test_gemm_mx_convt3(hip8, emu, b, t, hw, c):
    x = rnd(b * t * hw, c, seed=20)
    wt = rnd(c, 3 * c, seed=21, scale=(3 * c) ** -0.5)
    bias = rnd(c, seed=22, dtype=torch.float32)
    res = rnd(b * t * hw, c, seed=23)
    geom = dict(kind="t3", frames=b * t, t_len=t, cin=c, h_out=1, w_out=hw)
    check(hip8.gemm(x, wt, bias, conv=geom, residual=res, out_scale=0.3),
          emu.gemm(fq(x), fq(wt), bias, conv=geom, residual=res, out_scale=0.3), f"mx convt3 b{b} t{t} hw{hw} c{c}")


def test_mx_bad_arguments_raise(hip8):
    from tooncrafter_amd._lib import TooncrafterHipError
    with pytest.raises(ValueError):
        hip8.quant_mxfp8(rnd(8, 48), 48)                       # k not a multiple of 32
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip8.quant_mxfp8(torch.zeros(8, 64, dtype=BF16), 64)   # CPU tensor


def test_ineligible_shapes_stay_on_the_bf16_kernel(hip8, emu):
    a, w = rnd(1200, 72, seed=1), rnd(64, 72, seed=2)          # K not a multiple of 32
    c0 = dict(hip8.fp8_calls)
    check(hip8.gemm(a, w), emu.gemm(a, w), "bf16 fallthrough")
    assert hip8.fp8_calls["bf16"] == c0["bf16"] + 1 and hip8.fp8_calls["mx"] == c0["mx"]


@pytest.mark.parametrize("rows,c", [(1000, 320), (77, 640), (130, 1280), (33, 960), (64, 512), (5, 2048)])
def test_layernorm_emitting_mxfp8_is_layernorm_then_quantiser(hip8, rows, c):
    """tc_layernorm_mxfp8 (the quantiser fused into its producer) == tc_quant_mxfp8(tc_layernorm(x)), bit for bit,
    scale padding included; and the GEMM fed with it == the GEMM fed with the bf16 rows."""
    from tooncrafter_amd.ops import MxRows
    x = rnd(rows, c, seed=50, scale=3.0)
    g, b = rnd(c, seed=51, dtype=torch.float32), rnd(c, seed=52, dtype=torch.float32)
    y = hip8.layernorm(x, g, b)
    q_ref, s_ref = hip8.quant_mxfp8(y)
    mx_rows = hip8.layernorm(x, g, b, mx_for=(4 * c, 4 * c))
    assert isinstance(mx_rows, MxRows) and mx_rows.shape == (rows, c)
    assert torch.equal(mx_rows.q, q_ref) and torch.equal(mx_rows.scales, s_ref)
    w = rnd(4 * c, c, seed=53, scale=c ** -0.5)
    c0 = dict(hip8.fp8_calls)
    fused = hip8.gemm(mx_rows, w)
    ran_mx(hip8, c0)
    assert torch.equal(fused, hip8.gemm(y, w))


def test_layernorm_keeps_bf16_when_its_consumer_is_not_routed(hip8):
    x = rnd(64, 320, seed=54)
    g, b = rnd(320, seed=55, dtype=torch.float32), rnd(320, seed=56, dtype=torch.float32)
    old = hip8.fp8_min_n
    hip8.fp8_min_n = 4096
    try:
        y = hip8.layernorm(x, g, b, mx_for=(960, 960))
    finally:
        hip8.fp8_min_n = old
    assert isinstance(y, torch.Tensor) and y.dtype == BF16
