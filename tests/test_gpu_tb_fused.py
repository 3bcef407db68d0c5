"""The temporal self-attention of a level-0 transformer block as ONE launch (ABI 9: tc_temporal_attn_fused,
csrc/tb_fused.hip; reference lvdm/modules/attention.py:81-144 over the 16 frames of a pixel, TemporalTransformer
attention.py:365-412 behind norm1 / norm2, attention.py:225-246).

Checked against (a) the launches it replaces -- tc_layernorm, tc_gemm_bf16 (qkv), tc_attn_temporal, tc_gemm_bf16 (+residual);
(b) the emulated operator; (c) the fp64 statement of the reference's block.  Shapes: the BASELINE level-0 geometry
(B = 2, 16 frames, 40 x 64 pixels), one tile, several tiles per block with a ragged block count, rows at a wider pitch,
with and without the LayerNorm; repeated launches are bit-identical.
"""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd
from tooncrafter_amd.lvdm.common import fold_layernorm, pack_linear

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
C, HEADS, T = 320, 5, 16


@pytest.fixture(autouse=True)
def _fused_route_on(monkeypatch):
    """Since round 6 the level-0 default is the three-launch chain around csrc/qkv_attn.hip (ahead of this kernel in the
    forward-level A/B, profiles/r06_l0_chain_vs_tb_fused_forward_ab*.txt); the kernel stays in the library behind
    TC_TB_FUSED=1 and stays tested."""
    monkeypatch.setenv("TC_TB_FUSED", "1")


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def weights():
    wq, wk, wv = (rnd(C, C, seed=s, scale=0.06, dtype=torch.float32) for s in (1, 2, 3))
    wo, bo = rnd(C, C, seed=4, scale=0.05, dtype=torch.float32), rnd(C, seed=5, scale=0.1, dtype=torch.float32)
    gamma = rnd(C, seed=6, scale=0.2, dtype=torch.float32) + 1.0
    beta = rnd(C, seed=7, scale=0.1, dtype=torch.float32)
    raw = torch.cat([wq, wk, wv], 0)
    wf, bf = fold_layernorm(raw, None, gamma, beta)
    return dict(raw=(raw, wo, bo, gamma, beta), folded=(pack_linear(wf), bf.contiguous()),
                plain=(pack_linear(raw), torch.zeros(3 * C, device=raw.device)), wo=pack_linear(wo), bo=bo.contiguous())


def _x(b, hw, seed=11, pitch=C):
    full = rnd(b * T * hw, pitch, seed=seed, scale=1.5) + 0.3
    return full.to(BF16)[:, :C]


def _chain(hip, x, wts, ln, b, hw):
    xc = x.contiguous()
    if ln:
        ones, zeros = torch.ones(C, device=x.device), torch.zeros(C, device=x.device)
        qkv = hip.gemm(hip.layernorm(xc, ones, zeros, 1e-5), *wts["folded"])
    else:
        qkv = hip.gemm(xc, wts["plain"][0])
    a = hip.attention_temporal(qkv, b=b, t=T, hw=hw, heads=HEADS)
    return hip.gemm(a, wts["wo"], wts["bo"], residual=xc)


CASES = [("level-0 geometry (B = 2, 40 x 64)", 2, 2560, C), ("one tile (8 pixels)", 1, 8, C), ("three tiles", 1, 24, C),
         ("ragged block count (B = 3, 2 x 300 + 8 pixels)", 3, 608, C), ("rows at pitch 960", 1, 256, 960)]


@pytest.mark.parametrize("ln", [True, False], ids=["layernorm", "plain"])
@pytest.mark.parametrize("tag,b,hw,pitch", CASES, ids=[c[0] for c in CASES])
def test_fused_vs_chain_and_emulation(hip, weights, tag, b, hw, pitch, ln):
    x = _x(b, hw, pitch=pitch)
    assert hip.temporal_attn_fused_eligible(b=b, t=T, hw=hw, c=C, heads=HEADS, ldx=x.stride(0))
    w, bias = weights["folded"] if ln else weights["plain"]
    kw = dict(b=b, t=T, hw=hw, heads=HEADS, ln_eps=1e-5 if ln else None)
    out = hip.temporal_attn_fused(x, w, bias, weights["wo"], weights["bo"], **kw)
    torch.cuda.synchronize()
    ref = _chain(hip, x, weights, ln, b, hw)
    d = (out.float() - ref.float()).abs()
    scale = float(ref.float().abs().max())
    print(f"{tag} / {'LN' if ln else 'plain'}: fused vs four launches: max |d| {float(d.max()):.3e} "
          f"({float(d.max()) / (scale * 2 ** -8):.2f} bf16-ulp of scale), {int((d > 0).sum())} of {d.numel()} elements differ")
    check(out, ref, f"{tag}: fused vs the four launches", rel=3e-3)
    if b * T * hw <= 8192:
        emu = EmuOps(round_bf16=True, tb_fused_c=C)
        want = emu.temporal_attn_fused(x.cpu(), w.cpu(), bias.cpu(), weights["wo"].cpu(), weights["bo"].cpu(), **kw)
        check(out.cpu(), want, f"{tag}: fused vs emulation")
    again = hip.temporal_attn_fused(x, w, bias, weights["wo"], weights["bo"], **kw)
    assert torch.equal(out, again), "repeated launch differs"


def test_fused_vs_fp64_reference_block(hip, weights):
    """x + to_out(softmax(q k^T / 8) v) over the frames of every pixel, q / k / v = Linear(LayerNorm(x)), in fp64."""
    raw, wo, bo, gamma, beta = (t.double().cpu() for t in weights["raw"])
    b, hw = 1, 64
    x = _x(b, hw, seed=21)
    out = hip.temporal_attn_fused(x, *weights["folded"], weights["wo"], weights["bo"], b=b, t=T, hw=hw, heads=HEADS,
                                  ln_eps=1e-5).double().cpu()
    xd = x.double().cpu()
    qkv = torch.nn.functional.layer_norm(xd, (C,), gamma, beta, 1e-5) @ raw.t()
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(b, T, hw, HEADS, 64).permute(0, 2, 3, 1, 4) for i in range(3))
    o = ((q @ k.transpose(-1, -2)) * 64 ** -0.5).softmax(-1) @ v                       # [b, hw, heads, T, 64]
    ref = xd + o.permute(0, 3, 1, 2, 4).reshape(b * T * hw, C) @ wo.t() + bo
    err = float((out - ref).norm() / ref.norm())
    print(f"fused temporal attention vs fp64 reference block: rel-L2 {err:.3e}")
    assert err < 6e-3


def test_attention_is_over_frames_of_the_same_pixel(hip, weights):
    """Changing ONE pixel's rows changes that pixel's 16 output rows and nothing else (gather / tile-row mapping)."""
    b, hw = 1, 40
    x = _x(b, hw, seed=31).contiguous()
    kw = dict(b=b, t=T, hw=hw, heads=HEADS, ln_eps=1e-5)
    y0 = hip.temporal_attn_fused(x, *weights["folded"], weights["wo"], weights["bo"], **kw)
    x2 = x.clone()
    pix = 13
    rows = torch.arange(T, device=x.device) * hw + pix
    x2[rows[5]] = (x2[rows[5]].float() * -0.7 + 0.2).to(BF16)                         # frame 5 of pixel 13
    y1 = hip.temporal_attn_fused(x2, *weights["folded"], weights["wo"], weights["bo"], **kw)
    changed = (y0 != y1).any(dim=1).nonzero().flatten().tolist()
    assert set(changed) <= set(rows.tolist()) and len(changed) >= T - 1, changed


def test_eligibility_and_refusals(hip, weights, monkeypatch):
    from tooncrafter_amd._lib import TooncrafterHipError
    assert not hip.temporal_attn_fused_eligible(b=2, t=16, hw=640, c=640, heads=10)
    assert not hip.temporal_attn_fused_eligible(b=2, t=8, hw=2560, c=320, heads=5)
    assert not hip.temporal_attn_fused_eligible(b=2, t=16, hw=2556, c=320, heads=5)        # hw % 8
    x = _x(1, 8)
    with pytest.raises(ValueError):
        hip.temporal_attn_fused(x, weights["wo"], weights["folded"][1], weights["wo"], weights["bo"], b=1, t=T, hw=8, heads=HEADS)
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.temporal_attn_fused(x.cpu(), *weights["folded"], weights["wo"], weights["bo"], b=1, t=T, hw=8, heads=HEADS)
    monkeypatch.setenv("TC_TB_FUSED", "0")
    assert not hip.temporal_attn_fused_eligible(b=2, t=16, hw=2560, c=320, heads=5)


def test_block_routes_through_the_fused_operator(hip, monkeypatch):
    """A level-0 BasicTransformerBlock (temporal flavour) on the HIP backend, fused temporal attention on vs off."""
    from tooncrafter_amd import ops
    from tooncrafter_amd.lvdm.attention import BasicTransformerBlock
    from tooncrafter_amd.lvdm.common import Act
    torch.manual_seed(0)
    blk = BasicTransformerBlock(320, 5, 64, context_dim=None).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.05)
        for i in (1, 2, 3):
            getattr(blk, f"norm{i}").weight.add_(1.0)
    blk = blk.cuda()
    prev = ops.set_backend(hip)
    try:
        b, t, h, w = 1, 16, 8, 16
        x = rnd(b * t * h * w, 320, seed=41)
        act = Act(x, b, t, h, w)
        calls = []
        real = hip.temporal_attn_fused
        monkeypatch.setattr(hip, "temporal_attn_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            y1 = blk.forward_temporal(x, act)
            monkeypatch.setenv("TC_TB_FUSED", "0")
            y0 = blk.forward_temporal(x, act)
        assert len(calls) == 2
        check(y1, y0, "temporal block, fused temporal attention on vs off", rel=8e-3)
        monkeypatch.delenv("TC_TB_FUSED")                         # the default: not taken
        n = len(calls)
        with torch.no_grad():
            blk.forward_temporal(x, act)
        assert len(calls) == n
    finally:
        ops.set_backend(prev)
