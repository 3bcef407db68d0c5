"""GroupNorm(+SiLU) folded into the convolution that consumes it (ABI 10: tc_groupnorm_scale_shift + tc_conv_gn_bf16,
csrc/conv_halo.hip; reference lvdm/basics.py:76-87 in front of lvdm/modules/networks/openaimodel3d.py:154,179,255-266): the
HOST half on the CPU emulation -- ResBlock / TemporalConvBlock call ONE operator per (norm, activation, convolution)
triple (`ops.gn_conv`), which takes the one-pass route where the library's shape rule allows and the two launches
elsewhere -- and the operator's own contract: act(x * scale + shift) rounded to bf16 IS what tc_groupnorm would have
stored, and the zero padding applies to the activation."""
import pytest
import torch

from emu_ops import EmuOps
from tooncrafter_amd import ops
from tooncrafter_amd.lvdm.common import Act
from tooncrafter_amd.lvdm.openaimodel3d import ResBlock


def _resblock(c):
    torch.manual_seed(0)
    blk = ResBlock(c, 128, 0.0, out_channels=c, use_temporal_conv=True).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.03)
        for m in blk.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.add_(1.0)
    blk.emb_slice = (0, c)
    return blk


def _run(fuse, c, b, t, h, w):
    blk = _resblock(c)
    emu = EmuOps(round_bf16=True, gn_fuse=fuse)
    prev = ops.set_backend(emu)
    try:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(b * t * h * w, c, generator=g) + 0.5).to(torch.bfloat16)
        emb = torch.randn(b, c, generator=g)
        with torch.no_grad():
            y = blk(Act(x, b, t, h, w), emb).rows.float()
    finally:
        ops.set_backend(prev)
    return y, emu


def test_resblock_takes_the_one_pass_route_for_all_six_norms():
    """320 channels, 16 frames of 10 x 16 pixels: every convolution of the block tiles into patches (level 2 of the UNet)."""
    y0, e0 = _run(False, 320, 1, 16, 10, 16)
    y1, e1 = _run(True, 320, 1, 16, 10, 16)
    assert e0.gn_fuse_calls == {"fused": 0, "separate": 6}
    assert e1.gn_fuse_calls == {"fused": 6, "separate": 0}      # in_layers, out_layers, four temporal convolutions
    rel = float((y1 - y0).norm() / y0.norm())
    print("ResBlock with GroupNorm inside its convolutions vs two launches: rel-L2", rel)
    assert rel < 5e-3                   # fp64 E[x^2] - mean^2 vs F.group_norm: bf16 rounding flips only


def test_shapes_outside_the_rule_keep_the_two_launches():
    """8 x 8 images do not tile into 10 x 16 patches; 4 frames are no 16-frame clip: nothing changes, bit for bit."""
    y0, e0 = _run(False, 64, 2, 4, 8, 8)
    y1, e1 = _run(True, 64, 2, 4, 8, 8)
    assert e1.gn_fuse_calls == {"fused": 0, "separate": 6}
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("kind", ["3x3", "t3"])
def test_gn_conv_equals_groupnorm_then_convolution(kind):
    """The operator against its two-launch statement, incl. the epilogue arguments and the zero padding of the ACTIVATION:
    with beta far from zero, silu(shift) is not small -- a kernel that normalised the padding would be far off."""
    torch.manual_seed(3)
    c, n, frames, h, w = 64, 160, 16, 10, 16
    m = frames * h * w
    x = (torch.randn(m, c) * 2 + 1).to(torch.bfloat16)
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) + 3.0
    if kind == "3x3":
        conv = dict(kind="3x3", frames=frames, cin=c, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
        wt = (torch.randn(n, 9 * c) * (9 * c) ** -0.5).to(torch.bfloat16)
        gn = dict(samples=frames, rows=h * w)
    else:
        conv = dict(kind="t3", frames=frames, t_len=16, cin=c, h_out=h, w_out=w)
        wt = (torch.randn(n, 3 * c) * (3 * c) ** -0.5).to(torch.bfloat16)
        gn = dict(samples=1, rows=m)
    bias, res = torch.randn(n), torch.randn(m, n).to(torch.bfloat16)
    fused, plain = EmuOps(round_bf16=True, gn_fuse=True), EmuOps(round_bf16=True)
    a = fused.gn_conv(x, gamma, beta, wt, bias, eps=1e-5, conv=conv, residual=res, **gn)
    b = plain.gn_conv(x, gamma, beta, wt, bias, eps=1e-5, conv=conv, residual=res, **gn)
    assert fused.gn_fuse_calls["fused"] == 1 and plain.gn_fuse_calls["separate"] == 1
    rel = float((a.float() - b.float()).norm() / b.float().norm())
    assert rel < 3e-3, rel
    # the statistics table itself
    ss = fused.groupnorm_scale_shift(x, gamma, beta, eps=1e-5, **gn)
    y = (x.float().reshape(gn["samples"], gn["rows"], c) * ss[:, 0][:, None] + ss[:, 1][:, None]).reshape(m, c)
    ref = plain.groupnorm(x, gamma, beta, eps=1e-5, silu=False, **gn).float()
    assert float((y - ref).abs().max()) < 3e-2 * float(ref.abs().max())


def test_library_eligibility_rule_equals_the_emulation_rule():
    """tc_conv_gn_eligible is host logic: callable without a GPU.  The emulation's restatement (EmuOps.gn_conv_eligible), which
    the module-level tests above rely on, must agree with it on every shape class of the UNet and on the near misses."""
    import ctypes as C

    from tooncrafter_amd import _lib
    from tooncrafter_amd._lib import GATHER_CONV3x3, GATHER_CONVT3, TcGemmParams
    lib = _lib.load()
    emu = EmuOps(gn_fuse=True)
    cases = []
    for frames in (16, 32, 4):
        for h, w in ((40, 64), (20, 32), (10, 16), (5, 8), (8, 8), (30, 48), (10, 24)):
            for cin, n in ((320, 320), (640, 320), (64, 160), (320, 128), (96, 160)):
                for kind in ("3x3", "t3"):
                    for rows in (h * w, 16 * h * w, frames * h * w, 7):
                        cases.append((kind, frames, h, w, cin, n, rows))
    agree_yes = 0
    for kind, frames, h, w, cin, n, rows in cases:
        taps = 9 if kind == "3x3" else 3
        p = TcGemmParams()
        p.gather = GATHER_CONV3x3 if kind == "3x3" else GATHER_CONVT3
        p.cin, p.frames, p.t_len, p.h_out, p.w_out, p.h_in, p.w_in = cin, frames, 16 if kind == "t3" else 1, h, w, h, w
        p.stride, p.upsample, p.pad = 1, 0, 1
        p.m, p.n, p.k = frames * h * w, n, taps * cin
        p.lda, p.ldw, p.ldc, p.ldr, p.ldrb, p.row_div, p.alpha, p.out_scale, p.batch = cin, taps * cin, n, n, n, 1, 1.0, 1.0, 1
        got = bool(lib.tc_conv_gn_eligible(C.byref(p), rows))
        conv = dict(kind=kind, frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False, t_len=16)
        x = torch.empty((1, cin), dtype=torch.bfloat16)
        wt = torch.empty((n, taps * cin), dtype=torch.bfloat16)
        want = emu.gn_conv_eligible(x, wt, conv, rows)
        assert got == want, (kind, frames, h, w, cin, n, rows, got, want)
        agree_yes += got
    assert agree_yes >= 40           # the UNet's levels 0-2, per-frame and clip-wide statistics, both convolution kinds


def test_conv_gn_entry_refuses_bad_calls_before_it_launches():
    """Error paths of tc_conv_gn_bf16 / tc_groupnorm_scale_shift return before any launch: checkable without a GPU."""
    import ctypes as C

    from tooncrafter_amd import _lib
    from tooncrafter_amd._lib import GATHER_CONV3x3, TcGemmParams
    lib = _lib.load()
    EINVAL, EALIGN, ESHAPE, EWORKSPACE = -1, -2, -3, -4          # include/tooncrafter_hip.h (TcStatus); _lib.ERRORS
    buf = (C.c_char * 4096)()
    base = (C.addressof(buf) + 63) & ~63                     # a 64-byte aligned host address: never dereferenced on these paths
    p = TcGemmParams()
    p.gather, p.cin, p.frames, p.t_len, p.h_out, p.w_out, p.h_in, p.w_in = GATHER_CONV3x3, 64, 2, 1, 8, 8, 8, 8
    p.stride, p.upsample, p.pad = 1, 0, 1
    p.m, p.n, p.k = 2 * 64, 160, 9 * 64
    p.lda, p.ldw, p.ldc, p.ldr, p.ldrb, p.row_div, p.alpha, p.out_scale, p.batch = 64, 9 * 64, 160, 160, 160, 1, 1.0, 1.0, 1
    p.a = p.w = p.c = base
    assert lib.tc_conv_gn_bf16(C.byref(p), None, 64, 1, None) == EINVAL              # no table
    assert lib.tc_conv_gn_bf16(C.byref(p), base + 4, 64, 1, None) == EALIGN          # misaligned table
    assert lib.tc_conv_gn_bf16(C.byref(p), base, 64, 1, None) == ESHAPE              # 8 x 8 images: no patches
    p.a = 0
    assert lib.tc_conv_gn_bf16(C.byref(p), base, 64, 1, None) == EINVAL
    assert lib.tc_groupnorm_scale_shift(None, base, base, 2, 64, 64, 1e-5, base, base, 1 << 20, None) == EINVAL
    assert lib.tc_groupnorm_scale_shift(base, base, base, 2, 64, 48, 1e-5, base, base, 1 << 20, None) == ESHAPE   # C % 32
    assert lib.tc_groupnorm_scale_shift(base, base, base, 2, 64, 64, 1e-5, base, base, 16, None) == EWORKSPACE


def test_strict_halo_mode_refuses_a_convolution_it_cannot_take(monkeypatch):
    """TC_CONV_HALO=2 (the parity tests' mode): tc_gemm_bf16 fails with TC_ESHAPE -- before any launch, so checkable here --
    instead of silently falling back to the implicit GEMM for a convolution the tap-reuse kernel cannot take."""
    import ctypes as C

    from tooncrafter_amd import _lib
    from tooncrafter_amd._lib import GATHER_CONV3x3, TcGemmParams
    lib = _lib.load()
    buf = (C.c_char * 4096)()
    base = (C.addressof(buf) + 63) & ~63
    p = TcGemmParams()
    p.gather, p.cin, p.frames, p.t_len, p.h_out, p.w_out, p.h_in, p.w_in = GATHER_CONV3x3, 64, 2, 1, 17, 23, 17, 23
    p.stride, p.upsample, p.pad = 1, 0, 1
    p.m, p.n, p.k = 2 * 17 * 23, 160, 9 * 64
    p.lda, p.ldw, p.ldc, p.ldr, p.ldrb, p.row_div, p.alpha, p.out_scale, p.batch = 64, 9 * 64, 160, 160, 160, 1, 1.0, 1.0, 1
    p.a = p.w = p.c = base
    monkeypatch.setenv("TC_CONV_HALO", "2")
    assert lib.tc_gemm_bf16(C.byref(p), None) == -3                 # TC_ESHAPE: 17 x 23 images do not tile into patches
    assert lib.tc_gemm_gn_rows(C.byref(p)) == 0                     # and such a problem emits no producer statistics either
