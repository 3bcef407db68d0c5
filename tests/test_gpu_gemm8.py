"""8-wave persistent ping-pong GEMM (csrc/gemm8.hip; reference call sites: nn.Linear / nn.Conv2d 3x3 / nn.Conv3d (3,1,1)
of lvdm/modules/attention.py:415-442, lvdm/modules/networks/openaimodel3d.py:154,179,255-266 and
lvdm/models/autoencoder_dualref.py:52-61) against the fp32 statement of the operator (tests/emu_ops.py) and against the
tile families it can replace.

The kernel's hazards are all in its hand-counted LDS-DMA pipeline (requests issued from inline asm, counted vmcnt, raw
barriers, two wave groups one barrier apart) and in the K-tile stream running across tile boundaries, so the cases
force: every gather mode, odd and even K-tile counts (the LDS buffer parity flips between tiles), two K-tiles only,
ragged M / N / K tails, more tiles than blocks (TC_G8_GRID=8: every block walks a sequence, with and without a next
tile), a batch (blockIdx.z), GEGLU / residual / row-bias epilogues -- and, because the K order per accumulator is the
same as in the 4-wave kernels, demand BIT-identical results to the default routing.
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


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


class env:
    """The library reads its tuning switches per call."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _both(hip, fn, grid=None):
    """fn() under the forced 8-wave kernel and under the routing without it; single-slice summation order only."""
    kv = dict(TC_GEMM8=2, TC_GEMM_SPLITK=0)
    if grid:
        kv["TC_G8_GRID"] = grid
    with env(**kv):
        new = fn()
    with env(TC_GEMM8=0, TC_GEMM_SPLITK=0):
        old = fn()
    torch.cuda.synchronize()
    return new, old


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (300, 256, 192), (1000, 520, 392), (4096, 640, 2560), (2048, 2048, 2048),
                                   (5120, 1280, 320), (777, 264, 1280)])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("grid", [None, 8])
def test_linear(hip, emu, m, n, k, res, grid):
    a, w, bias = rnd(m, k, seed=m + 1), rnd(n, k, seed=n + 2, scale=k ** -0.5), rnd(n, seed=3, dtype=torch.float32)
    residual = rnd(m, n, seed=4) if res else None
    new, old = _both(hip, lambda: hip.gemm(a, w, bias, residual=residual), grid)
    check(new, emu.gemm(a, w, bias, residual=residual), f"gemm8 linear {m}x{n}x{k} res={res} grid={grid}")
    assert torch.equal(new, old), "8-wave kernel and 4-wave kernels differ bit-wise on the same summation order"


@pytest.mark.parametrize("m,n,k", [(512, 512, 128), (2560, 2560, 320), (1300, 640, 640)])
@pytest.mark.parametrize("grid", [None, 8])
def test_geglu(hip, emu, m, n, k, grid):
    from tooncrafter_amd.lvdm.common import pack_geglu
    a = rnd(m, k, seed=11)
    w, bias = pack_geglu(rnd(n, k, seed=12, scale=k ** -0.5, dtype=torch.float32), rnd(n, seed=13, dtype=torch.float32))
    new, old = _both(hip, lambda: hip.gemm(a, w, bias, act=ACT_GEGLU), grid)
    check(new, emu.gemm(a, w, bias, act=ACT_GEGLU), f"gemm8 GEGLU {m}x{n}x{k} grid={grid}")
    assert torch.equal(new, old)


@pytest.mark.parametrize("frames,h,w_,cin,n", [(3, 17, 23, 64, 264), (2, 40, 64, 320, 320), (4, 16, 16, 128, 512), (1, 9, 300, 192, 256)])
@pytest.mark.parametrize("grid", [None, 8])
def test_conv3x3(hip, emu, frames, h, w_, cin, n, grid):
    conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    m = frames * h * w_
    a = rnd(m, cin, seed=21)
    w, bias = rnd(n, 9 * cin, seed=22, scale=(9 * cin) ** -0.5), rnd(n, seed=23, dtype=torch.float32)
    rb, residual = rnd(frames, n, seed=24, dtype=torch.float32), rnd(m, n, seed=25)
    kw = dict(conv=conv, row_bias=rb, row_div=h * w_, residual=residual)
    new, old = _both(hip, lambda: hip.gemm(a, w, bias, **kw), grid)
    check(new, emu.gemm(a, w, bias, **kw), f"gemm8 conv3x3 {frames}x{h}x{w_} {cin}->{n} grid={grid}")
    assert torch.equal(new, old)


@pytest.mark.parametrize("frames,hw,cin,n", [(32, 70, 128, 136), (16, 640, 320, 320), (48, 25, 256, 512)])
@pytest.mark.parametrize("grid", [None, 8])
def test_conv_t3(hip, emu, frames, hw, cin, n, grid):
    conv = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=1, w_out=hw)
    m = frames * hw
    a = rnd(m, cin, seed=31)
    w, bias = rnd(n, 3 * cin, seed=32, scale=(3 * cin) ** -0.5), rnd(n, seed=33, dtype=torch.float32)
    residual = rnd(m, n, seed=34)
    new, old = _both(hip, lambda: hip.gemm(a, w, bias, conv=conv, residual=residual), grid)
    check(new, emu.gemm(a, w, bias, conv=conv, residual=residual), f"gemm8 convT3 {frames}x{hw} {cin}->{n} grid={grid}")
    assert torch.equal(new, old)


def test_strided_views_and_untouched_neighbours(hip, emu):
    """A, C and the residual as column slices of wider buffers; nothing outside C may be written (ragged last tiles)."""
    m, n, k = 1100, 328, 448
    a = rnd(m, 3 * k, seed=41)[:, k:2 * k]
    w, bias = rnd(n, k, seed=42, scale=k ** -0.5), rnd(n, seed=43, dtype=torch.float32)
    res = rnd(m, 2 * n, seed=44)[:, n:]
    outbuf = torch.full((m + 300, 3 * n), 7.0, dtype=BF16, device=DEV)
    with env(TC_GEMM8=2, TC_G8_GRID=8):
        hip.gemm(a, w, bias, residual=res, out=outbuf[:m, n:2 * n])
    check(outbuf[:m, n:2 * n], emu.gemm(a, w, bias, residual=res), "gemm8 strided A / C / residual")
    assert float((outbuf[:m, :n] - 7).abs().max()) == 0 and float((outbuf[:m, 2 * n:] - 7).abs().max()) == 0
    assert float((outbuf[m:] - 7).abs().max()) == 0, "rows behind M were written"


def test_transpose_detecting(hip):
    """Identity-like A against an asymmetric W: a swapped fragment / C-write mapping cannot pass."""
    m, n, k = 512, 512, 512
    a = torch.eye(m, k, device=DEV, dtype=BF16)
    w = ((torch.arange(n, device=DEV)[:, None] * 3 + torch.arange(k, device=DEV)[None, :] % 7).float())
    w = (w / w.max()).to(BF16)
    with env(TC_GEMM8=2):
        out = hip.gemm(a, w)
    assert torch.equal(out, w.t().contiguous()), "fragment or C-write layout is wrong"


def test_repeated_launches_are_bit_identical(hip):
    """Race screen: 30 launches of a many-tile problem under a small grid (long tile walks), all identical."""
    m, n, k = 8192, 1280, 960
    a, w = rnd(m, k, seed=51), rnd(n, k, seed=52, scale=k ** -0.5)
    res = rnd(m, n, seed=53)
    with env(TC_GEMM8=2, TC_G8_GRID=24):
        first = hip.gemm(a, w, residual=res)
        for _ in range(30):
            assert torch.equal(hip.gemm(a, w, residual=res), first), "a launch differs: a tile was read before it landed"


def test_default_heuristic_takes_the_level1_ff2(hip, emu):
    """The routing rule of tc_gemm8_try (one round of tiles, long K): same bits with and without it."""
    m, n, k = 20480, 640, 2560
    a, w, bias = rnd(m, k, seed=61), rnd(n, k, seed=62, scale=k ** -0.5), rnd(n, seed=63, dtype=torch.float32)
    res = rnd(m, n, seed=64)
    with env(TC_GEMM8=1):
        new = hip.gemm(a, w, bias, residual=res)
    with env(TC_GEMM8=0):
        old = hip.gemm(a, w, bias, residual=res)
    assert torch.equal(new, old)
    check(new, emu.gemm(a, w, bias, residual=res), "gemm8 level-1 ff2")
