"""The 160x160 GEMM's K loop with the tile requests BETWEEN the MFMAs (csrc/gemm16.hip: TC_G16_ILV = 1 | 2) and its tall
320 x 160 tile on eight waves (TC_G16_TALL; the third W pass half empty: its four pieces go to a dump area) -- reference call
sites: nn.Linear / nn.Conv2d 3x3 / nn.Conv3d (3,1,1) of lvdm/modules/networks/openaimodel3d.py:154,179,255-266 and
lvdm/modules/attention.py:415-442) against the fp32 statement of the operator (tests/emu_ops.py) and against the plain loop.

Both loops keep the plain loop's MFMA order per accumulator, so the results must be BIT-identical to it; what can go
wrong is in the hand-ordered pipeline -- requests from inline asm into the stage a barrier has just released, fragment
reads of the other stage beside them, one wait + barrier per K-step, in loop 2 a second fragment set read one K-slice
ahead -- so the cases force: every gather mode, one / two / odd / even K-tile counts, ragged M and K tails, zero-padded
image borders and clip ends, a batch (blockIdx.z), row-bias / residual / activation epilogues, strided operands."""
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


ARMS = [(0, 0), (1, 0), (2, 0), (0, 2), (1, 2), (2, 2)]     # (TC_G16_ILV, TC_G16_TALL): loop x tile height (160 | 320 rows, 8 waves)


def _arms(fn):
    """fn() on the 160-column kernel (forced) under the plain loop and both interleaved loops, on both tile heights."""
    out = {}
    for ilv, tall in ARMS:
        with env(TC_GEMM_TILE16=2, TC_G16_ILV=ilv, TC_G16_TALL=tall, TC_GEMM_SPLITK=0, TC_GEMM8=0, TC_GEMM_WS=0):
            out[(ilv, tall)] = fn()
    torch.cuda.synchronize()
    return out


def _same(out, ref, what):
    base = out[(0, 0)]
    check(base, ref, what)
    for arm in ARMS[1:]:
        assert torch.equal(out[arm], base), f"{what}: loop {arm[0]} / tall {arm[1]} differs bit-wise from the plain 160-row loop"


@pytest.mark.parametrize("m,n,k", [(160, 160, 64), (320, 160, 128), (1000, 320, 192), (4096, 640, 2560), (5120, 1280, 320),
                                   (777, 480, 1288), (20480, 640, 640), (333, 160, 40)])
@pytest.mark.parametrize("res", [False, True])
def test_linear(hip, emu, m, n, k, res):
    a, w, bias = rnd(m, k, seed=m + 1), rnd(n, k, seed=n + 2, scale=k ** -0.5), rnd(n, seed=3, dtype=torch.float32)
    residual = rnd(m, n, seed=4) if res else None
    _same(_arms(lambda: hip.gemm(a, w, bias, residual=residual, act=ACT_SILU if res else ACT_NONE)),
          emu.gemm(a, w, bias, residual=residual, act=ACT_SILU if res else ACT_NONE), f"gemm16 ilv linear {m}x{n}x{k} res={res}")


@pytest.mark.parametrize("frames,h,w_,cin,n", [(3, 17, 23, 64, 160), (2, 40, 64, 320, 320), (4, 16, 16, 128, 480), (1, 9, 300, 192, 160),
                                                 (32, 20, 32, 640, 640), (5, 5, 8, 1280, 320)])
def test_conv3x3(hip, emu, frames, h, w_, cin, n):
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    m = frames * h * w_
    a = rnd(m, cin, seed=21)
    w, bias = rnd(n, 9 * cin, seed=22, scale=(9 * cin) ** -0.5), rnd(n, seed=23, dtype=torch.float32)
    rb, residual = rnd(frames, n, seed=24, dtype=torch.float32), rnd(m, n, seed=25)
    kw = dict(conv=conv, row_bias=rb, row_div=h * w_, residual=residual)
    _same(_arms(lambda: hip.gemm(a, w, bias, **kw)), emu.gemm(a, w, bias, **kw), f"gemm16 ilv conv3x3 {frames}x{h}x{w_} {cin}->{n}")


def test_conv3x3_stride2_and_upsample_stay_on_the_plain_loop(hip, emu):
    """The interleaved loops carry the slim request state (stride-1 taps only): the strided and the nearest-x2 gathers must
    be routed to the plain loop whatever TC_G16_ILV says, and stay exact."""
    for stride, up in ((2, False), (1, True)):
        h, w_ = 16, 20
        ho, wo = (h * 2, w_ * 2) if up else ((h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1)
        conv = dict(kind="3x3", frames=3, cin=128, h_in=h, w_in=w_, h_out=ho, w_out=wo, stride=stride, upsample=up)
        a = rnd(3 * h * w_, 128, seed=41)
        w, bias = rnd(160, 9 * 128, seed=42, scale=(9 * 128) ** -0.5), rnd(160, seed=43, dtype=torch.float32)
        _same(_arms(lambda: hip.gemm(a, w, bias, conv=conv)), emu.gemm(a, w, bias, conv=conv), f"gemm16 ilv conv3x3 stride {stride} up {up}")


@pytest.mark.parametrize("frames,hw,cin,n", [(32, 70, 128, 160), (16, 640, 320, 320), (48, 25, 256, 480), (32, 160, 1280, 320)])
def test_conv_t3(hip, emu, frames, hw, cin, n):
    conv = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=1, w_out=hw)
    m = frames * hw
    a = rnd(m, cin, seed=31)
    w, bias = rnd(n, 3 * cin, seed=32, scale=(3 * cin) ** -0.5), rnd(n, seed=33, dtype=torch.float32)
    residual = rnd(m, n, seed=34)
    _same(_arms(lambda: hip.gemm(a, w, bias, conv=conv, residual=residual)), emu.gemm(a, w, bias, conv=conv, residual=residual),
          f"gemm16 ilv convT3 {frames}x{hw} {cin}->{n}")


def test_strided_views_and_untouched_neighbours(hip, emu):
    """A, C and the residual as column slices of wider buffers; nothing outside C may be written (ragged last tile)."""
    m, n, k = 1100, 320, 448
    a = rnd(m, 3 * k, seed=41)[:, k:2 * k]
    w, bias = rnd(n, k, seed=42, scale=k ** -0.5), rnd(n, seed=43, dtype=torch.float32)
    res = rnd(m, 2 * n, seed=44)[:, n:]
    outs = {}
    for ilv, tall in ARMS:
        outbuf = torch.full((m + 300, 3 * n), 7.0, dtype=BF16, device="cuda")
        with env(TC_GEMM_TILE16=2, TC_G16_ILV=ilv, TC_G16_TALL=tall, TC_GEMM_SPLITK=0, TC_GEMM8=0, TC_GEMM_WS=0):
            hip.gemm(a, w, bias, residual=res, out=outbuf[:m, n:2 * n])
        torch.cuda.synchronize()
        assert float((outbuf[:m, :n] - 7).abs().max()) == 0 and float((outbuf[:m, 2 * n:] - 7).abs().max()) == 0
        assert float((outbuf[m:] - 7).abs().max()) == 0, "rows behind M were written"
        outs[(ilv, tall)] = outbuf[:m, n:2 * n].clone()
    _same(outs, emu.gemm(a, w, bias, residual=res), "gemm16 ilv strided A / C / residual")


@pytest.mark.parametrize("ilv,tall", ARMS[1:])
def test_repeated_launches_are_bit_identical(hip, ilv, tall):
    """Race screen: 30 launches of a two-round problem (1024 / 512 tiles on 512 / 256 block slots), all identical."""
    frames, h, w_, cin, n = 32, 40, 64, 320, 320
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    a, w = rnd(frames * h * w_, cin, seed=61), rnd(n, 9 * cin, seed=62, scale=(9 * cin) ** -0.5)
    with env(TC_GEMM_TILE16=2, TC_G16_ILV=ilv, TC_G16_TALL=tall):
        first = hip.gemm(a, w, conv=conv)
        for _ in range(30):
            assert torch.equal(hip.gemm(a, w, conv=conv), first), "a launch differs: a tile was read before it landed"
