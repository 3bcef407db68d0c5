"""The tap-reuse convolution kernel (csrc/conv_halo.hip; the default route of the UNet's 3x3 convolutions at levels 0-2 since
round 5) -- reference call sites: nn.Conv2d 3x3 and nn.Conv3d (3,1,1) of lvdm/modules/networks/openaimodel3d.py:154,179,255-266
-- against the fp32 statement of the operator (tests/emu_ops.py) and against the implicit-GEMM kernels it replaces.

Written in round 4 without GPU access (index arithmetic on the CPU: tests/test_conv_halo_cpu.py); these tests were gated then
and passed on their first execution in round 5 (profiles/r05_pytest_conv_halo_first_gpu_run.log).  TC_CONV_HALO=2 is the
strict mode: tc_gemm_bf16 FAILS if a convolution is not taken by the kernel, so a passing case has provably run it (no silent
fallback); it also takes the temporal geometry, which the default routing leaves to the implicit GEMM (measured 0.5-0.99x)."""
import os

import pytest
import torch

from emu_ops import EmuOps
from test_gpu_gemm8 import env
from test_gpu_ops import check, rnd
from tooncrafter_amd._lib import ACT_NONE, ACT_SILU

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def _both(fn, tall_ok):
    """(160-row patches, 320-row patches or None, implicit GEMM).  The tall kernel (TC_CONV_HALO_TALL=2) takes a problem only
    if 20-row patches tile it; where they do not it falls back to the 160-row patches, which the first arm already covers."""
    with env(TC_CONV_HALO=2, TC_CONV_HALO_TALL=0, TC_CONV_HALO_KSPLIT=0):
        halo = fn()
    tall = None
    if tall_ok:
        with env(TC_CONV_HALO=2, TC_CONV_HALO_TALL=2, TC_CONV_HALO_KSPLIT=0):
            tall = fn()
    with env(TC_CONV_HALO=0):
        base = fn()
    torch.cuda.synchronize()
    return halo, tall, base


def _close(halo, tall, base, ref, what):
    check(base, ref, what + " (implicit GEMM)")
    check(halo, ref, what + " (halo patches)")
    # same operands, same fp32 products, another summation order: far inside a bf16 ulp of the result's scale
    d = (halo.float() - base.float()).abs().max().item()
    assert d <= 2.0 ** -6 * max(base.float().abs().max().item(), 1.0), f"{what}: halo vs implicit GEMM differ by {d}"
    if tall is not None:
        # a tall patch is two 160-row patches stacked: same chunk-major order per output element -> the same bits
        assert torch.equal(tall, halo), f"{what}: 320-row patches differ bit-wise from 160-row patches"


# (frames, h, w, cin, n): one patch; the UNet's levels 2 / 1 / 0 at B = 2; several patches per frame with every border kind
@pytest.mark.parametrize("frames,h,w_,cin,n", [(1, 10, 16, 64, 160), (3, 20, 32, 128, 320), (32, 10, 16, 1280, 1280),
                                                 (32, 20, 32, 640, 640), (32, 40, 64, 320, 320), (2, 40, 64, 960, 320)])
@pytest.mark.parametrize("epi", ["plain", "emb+res"])
def test_conv3x3(hip, emu, frames, h, w_, cin, n, epi):
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    m = frames * h * w_
    a = rnd(m, cin, seed=21)
    w, bias = rnd(n, 9 * cin, seed=22, scale=(9 * cin) ** -0.5), rnd(n, seed=23, dtype=torch.float32)
    kw = dict(conv=conv)
    if epi != "plain":
        kw.update(row_bias=rnd(frames, n, seed=24, dtype=torch.float32), row_div=h * w_, residual=rnd(m, n, seed=25), act=ACT_SILU)
    halo, tall, base = _both(lambda: hip.gemm(a, w, bias, **kw), h % 20 == 0)
    _close(halo, tall, base, emu.gemm(a, w, bias, **kw), f"conv3x3 {frames}x{h}x{w_} {cin}->{n} {epi}")


@pytest.mark.parametrize("frames,hw,cin,n", [(16, 10, 64, 160), (32, 160, 1280, 1280), (32, 640, 640, 640), (32, 2560, 320, 320),
                                              (48, 40, 128, 480)])
@pytest.mark.parametrize("res", [False, True])
def test_conv_t3(hip, emu, frames, hw, cin, n, res):
    conv = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=1, w_out=hw)
    m = frames * hw
    a = rnd(m, cin, seed=31)
    w, bias = rnd(n, 3 * cin, seed=32, scale=(3 * cin) ** -0.5), rnd(n, seed=33, dtype=torch.float32)
    residual = rnd(m, n, seed=34) if res else None
    halo, tall, base = _both(lambda: hip.gemm(a, w, bias, conv=conv, residual=residual), hw % 20 == 0)
    _close(halo, tall, base, emu.gemm(a, w, bias, conv=conv, residual=residual), f"convT3 {frames}x{hw} {cin}->{n} res={res}")


@pytest.mark.parametrize("kind,frames,h,w_,cin,n", [("3x3", 32, 10, 16, 1280, 1280), ("3x3", 2, 10, 16, 256, 160), ("3x3", 3, 20, 32, 640, 320),
                                                     ("t3", 32, 10, 16, 1280, 1280), ("t3", 16, 2, 5, 256, 160), ("t3", 32, 8, 20, 384, 320)])
def test_k_split_inside_the_block(hip, emu, kind, frames, h, w_, cin, n):
    """TC_CONV_HALO_KSPLIT: two 4-wave groups with half of the channel chunks each, accumulators handed over through LDS.
    0 = never, 2 = whenever cin / 64 is even (forced here also where the patches would fill the chip)."""
    if kind == "3x3":
        conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    else:
        conv = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w_)
    taps = 9 if kind == "3x3" else 3
    m = frames * h * w_
    a = rnd(m, cin, seed=81)
    w, bias = rnd(n, taps * cin, seed=82, scale=(taps * cin) ** -0.5), rnd(n, seed=83, dtype=torch.float32)
    residual = rnd(m, n, seed=84)
    out = {}
    for ks in (0, 2):
        with env(TC_CONV_HALO=2, TC_CONV_HALO_TALL=0, TC_CONV_HALO_KSPLIT=ks):
            out[ks] = hip.gemm(a, w, bias, conv=conv, residual=residual)
    torch.cuda.synchronize()
    ref = emu.gemm(a, w, bias, conv=conv, residual=residual)
    check(out[0], ref, f"halo {kind} {frames}x{h}x{w_} {cin}->{n}, one group")
    check(out[2], ref, f"halo {kind} {frames}x{h}x{w_} {cin}->{n}, K split over two groups")
    d = (out[2].float() - out[0].float()).abs().max().item()
    assert d <= 2.0 ** -6 * max(out[0].float().abs().max().item(), 1.0), f"K split vs one group differ by {d}"


def test_shapes_the_kernel_cannot_take_fall_through_in_mode_1_and_fail_in_mode_2(hip, emu):
    """17 x 23 images do not tile into 10 x 16 patches: mode 1 routes them to the implicit GEMM, strict mode refuses."""
    conv = dict(kind="3x3", frames=2, cin=64, h_in=17, w_in=23, h_out=17, w_out=23, stride=1, upsample=False)
    a, w = rnd(2 * 17 * 23, 64, seed=41), rnd(160, 9 * 64, seed=42, scale=(9 * 64) ** -0.5)
    with env(TC_CONV_HALO=1):
        got = hip.gemm(a, w, conv=conv)
    check(got, emu.gemm(a, w, conv=conv), "conv3x3 17x23 under TC_CONV_HALO=1")
    with env(TC_CONV_HALO=2):
        with pytest.raises(Exception):
            hip.gemm(a, w, conv=conv)
    torch.cuda.synchronize()


def test_strided_views_and_untouched_neighbours(hip, emu):
    """A, C and the residual as column slices of wider buffers; nothing outside C may be written."""
    frames, h, w_, cin, n = 2, 20, 16, 128, 160
    m = frames * h * w_
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    a = rnd(m, 3 * cin, seed=51)[:, cin:2 * cin]
    w, bias = rnd(n, 9 * cin, seed=52, scale=(9 * cin) ** -0.5), rnd(n, seed=53, dtype=torch.float32)
    res = rnd(m, 2 * n, seed=54)[:, n:]
    outbuf = torch.full((m + 300, 3 * n), 7.0, dtype=BF16, device="cuda")
    with env(TC_CONV_HALO=2):
        hip.gemm(a, w, bias, conv=conv, residual=res, out=outbuf[:m, n:2 * n])
    torch.cuda.synchronize()
    assert float((outbuf[:m, :n] - 7).abs().max()) == 0 and float((outbuf[:m, 2 * n:] - 7).abs().max()) == 0
    assert float((outbuf[m:] - 7).abs().max()) == 0, "rows behind M were written"
    check(outbuf[:m, n:2 * n], emu.gemm(a, w, bias, conv=conv, residual=res), "halo conv3x3, strided A / C / residual")


def test_locality_of_the_zero_padding(hip):
    """Changing ONE input pixel changes exactly its 3x3 neighbourhood of output pixels inside its own frame -- across a patch
    corner and at an image corner (a halo pixel read from the neighbouring row / frame would show up here)."""
    frames, h, w_, cin, n = 3, 20, 32, 64, 160
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    a = rnd(frames * h * w_, cin, seed=61)
    w = rnd(n, 9 * cin, seed=62, scale=(9 * cin) ** -0.5)
    with env(TC_CONV_HALO=2):
        base = hip.gemm(a, w, conv=conv).float()
        for f, y, x in ((1, 9, 15), (1, 10, 16), (0, 0, 0), (2, 19, 31), (1, 0, 31)):
            a2 = a.clone()
            a2[(f * h + y) * w_ + x] += 1.0
            d = (hip.gemm(a2, w, conv=conv).float() - base).abs().amax(1).view(frames, h, w_) > 0
            want = torch.zeros_like(d)
            want[f, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = True
            assert torch.equal(d, want), f"pixel ({f},{y},{x}): the changed outputs are not its 3x3 neighbourhood"
    torch.cuda.synchronize()


@pytest.mark.parametrize("tall", [0, 2])
def test_repeated_launches_are_bit_identical(hip, tall):
    """Race screen: 30 launches of the level-0 problem (1024 patches on 512 block slots | 512 tall ones on 256), all identical."""
    frames, h, w_, cin, n = 32, 40, 64, 320, 320
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    a, w = rnd(frames * h * w_, cin, seed=71), rnd(n, 9 * cin, seed=72, scale=(9 * cin) ** -0.5)
    with env(TC_CONV_HALO=2, TC_CONV_HALO_TALL=tall):
        first = hip.gemm(a, w, conv=conv)
        for _ in range(30):
            assert torch.equal(hip.gemm(a, w, conv=conv), first)
    torch.cuda.synchronize()


def test_default_routing_takes_3x3_and_leaves_temporal(hip, emu):
    """Mode 1 (no environment): a UNet 3x3 convolution runs on the halo kernel -- its result equals the strict mode's bit for
    bit and differs from the implicit GEMM's summation order -- and a temporal one stays on the implicit GEMM."""
    frames, h, w_, cin, n = 32, 20, 32, 640, 640
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    a, w = rnd(frames * h * w_, cin, seed=111), rnd(n, 9 * cin, seed=112, scale=(9 * cin) ** -0.5)
    for k in ("TC_CONV_HALO", "TC_CONV_HALO_T3", "TC_CONV_HALO_3X3", "TC_CONV_HALO_TALL", "TC_CONV_HALO_KSPLIT"):
        assert k not in os.environ, f"{k} is set: this test is about the default"
    dflt = hip.gemm(a, w, conv=conv)
    with env(TC_CONV_HALO=2):
        strict = hip.gemm(a, w, conv=conv)
    with env(TC_CONV_HALO=0):
        base = hip.gemm(a, w, conv=conv)
    assert torch.equal(dflt, strict) and not torch.equal(dflt, base)
    convt = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w_)
    wt = rnd(n, 3 * cin, seed=113, scale=(3 * cin) ** -0.5)
    dflt = hip.gemm(a, wt, conv=convt)
    with env(TC_CONV_HALO=0):
        base = hip.gemm(a, wt, conv=convt)
    assert torch.equal(dflt, base)
    check(dflt, emu.gemm(a, wt, conv=convt), "temporal convolution, default routing")
