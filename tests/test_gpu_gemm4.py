"""The four-wave 256x256 GEMM (csrc/gemm4.hip: one wave per SIMD, 128x128 outputs per wave, accumulators pinned to AGPRs,
MFMAs and tile requests from inline asm in a hand-laid order) -- reference call sites: nn.Linear / GEGLU of
lvdm/modules/attention.py:415-442 -- against the fp32 statement of the operator (tests/emu_ops.py) and against the tiled
kernels it competes with.  TC_GEMM4=2 takes every linear problem the kernel can run (0 = never; the default routes by the
measured rule)."""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_gemm8 import env
from test_gpu_ops import check, rnd
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU
from tooncrafter_amd.lvdm.common import pack_geglu

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


VARIANT = {"v": 2}          # TC_G4_VARIANT for the gemm4 arm (the `variant` fixture walks 0, 1, 2 over the linear cases)


def _both(fn):
    with env(TC_GEMM4=2, TC_G4_VARIANT=VARIANT["v"]):
        a = fn()
    with env(TC_GEMM4=0):
        b = fn()
    torch.cuda.synchronize()
    return a, b


def _same_within_rounding(new, base, what):
    # same products, fp32 accumulation in another MFMA shape / K order inside a step: a bf16 ulp on a few elements
    d = (new.float() - base.float()).abs().max().item()
    assert d <= 2.0 ** -6 * max(base.float().abs().max().item(), 1.0), f"{what}: differs from the tiled kernels by {d}"


# full tiles, ragged M and N (also below one tile), K tails (k % 64 != 0), K = 2 and 3 steps (prologue / tail loop only), long K
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 192), (1000, 520, 320), (77, 1280, 640), (5120, 1280, 5120),
                                   (300, 264, 136), (20480, 640, 2560), (257, 8, 4096), (2048, 2048, 2048), (600, 512, 384)])
@pytest.mark.parametrize("epi", ["plain", "bias+res", "silu+rowbias"])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_gemm4_linear(hip, emu, m, n, k, epi, variant):
    VARIANT["v"] = variant
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5)
    kw = {}
    bias = None
    if epi != "plain":
        bias = rnd(n, seed=3, dtype=torch.float32)
    if epi == "bias+res":
        kw["residual"] = rnd(m, n, seed=4)
    if epi == "silu+rowbias":
        kw.update(act=ACT_SILU, row_bias=rnd(4, n, seed=5, dtype=torch.float32), row_div=(m + 3) // 4, alpha=0.5, out_scale=1.5)
    new, base = _both(lambda: hip.gemm(a, w, bias, **kw))
    check(new, emu.gemm(a, w, bias, **kw), f"gemm4 {m}x{n}x{k} {epi}")
    _same_within_rounding(new, base, f"gemm4 {m}x{n}x{k} {epi}")
    VARIANT["v"] = 2


@pytest.mark.parametrize("m,n_out,k", [(20480, 2560, 640), (5120, 5120, 1280), (300, 64, 320), (256, 128, 128)])
def test_gemm4_geglu(hip, emu, m, n_out, k):
    """The GEGLU projections (weights packed per 32 columns: 16 values | 16 gates): out = (x W_v + b_v) * gelu(x W_g + b_g)."""
    g = torch.Generator().manual_seed(7)
    wp, bp = pack_geglu(torch.randn(2 * n_out, k, generator=g) * k ** -0.5, torch.randn(2 * n_out, generator=g) * 0.1)
    wp, bp = wp.to(DEV), bp.to(DEV)
    a = rnd(m, k, seed=8)
    new, base = _both(lambda: hip.gemm(a, wp, bp, act=ACT_GEGLU))
    assert new.shape == (m, n_out)
    check(new, emu.gemm(a, wp, bp, act=ACT_GEGLU), f"gemm4 GEGLU {m}x{n_out}x{k}")
    _same_within_rounding(new, base, f"gemm4 GEGLU {m}x{n_out}x{k}")


def test_gemm4_f32_output_batch_and_strided_views(hip, emu):
    a, w, b = rnd(600, 512, seed=11), rnd(520, 512, seed=12, scale=512 ** -0.5), rnd(520, seed=13, dtype=torch.float32)
    new, base = _both(lambda: hip.gemm(a, w, b, out_f32=True))
    assert new.dtype == torch.float32
    check(new, emu.gemm(a, w, b, out_f32=True), "gemm4 fp32 output", f32=True)
    # A and C as column slices of wider buffers; nothing outside C may be written
    wide = rnd(600, 3 * 512, seed=14)
    av = wide[:, 512:1024]
    outbuf = torch.full((700, 3 * 520), 7.0, dtype=torch.bfloat16, device=DEV)
    with env(TC_GEMM4=2):
        hip.gemm(av, w, b, out=outbuf[:600, 520:1040])
    torch.cuda.synchronize()
    assert float((outbuf[:600, :520] - 7).abs().max()) == 0 and float((outbuf[:600, 1040:] - 7).abs().max()) == 0
    assert float((outbuf[600:] - 7).abs().max()) == 0, "rows behind M were written"
    check(outbuf[:600, 520:1040], emu.gemm(av, w, b), "gemm4 strided A / C")
    # batched (grid.z): 3 independent products
    ab, wb = rnd(3 * 300, 256, seed=15), rnd(3 * 264, 256, seed=16, scale=256 ** -0.5)
    kwb = dict(batch=3, stride_a=300 * 256, stride_w=264 * 256, stride_c=300 * 264, m=300)
    ob = torch.empty((3 * 300, 264), dtype=torch.bfloat16, device=DEV)
    with env(TC_GEMM4=2):
        hip.gemm(ab[:300], wb[:264], out=ob[:300], **kwb)
    torch.cuda.synchronize()
    for i in range(3):
        check(ob[i * 300:(i + 1) * 300], emu.gemm(ab[i * 300:(i + 1) * 300], wb[i * 264:(i + 1) * 264]), f"gemm4 batch item {i}")


def test_gemm4_repeated_launches_are_bit_identical(hip):
    """Race screen for the hand-counted waits: 40 launches of a many-tile, long-K problem, all identical."""
    a, w = rnd(4096, 4096, seed=21), rnd(4096, 4096, seed=22, scale=4096 ** -0.5)
    for v in (0, 1, 2):
        with env(TC_GEMM4=2, TC_G4_VARIANT=v):
            first = hip.gemm(a, w)
            for _ in range(40):
                assert torch.equal(hip.gemm(a, w), first)
    torch.cuda.synchronize()
