"""The one-launch feed-forward (ABI 9: tc_ff_geglu_fused; reference lvdm/modules/attention.py:415-442 behind norm3,
attention.py:244-246): the host half on the CPU emulation -- BasicTransformerBlock._pre hands the RAW rows and the
LayerNorm-folded GEGLU weights to FeedForward, which takes the fused operator when the backend offers it."""
import torch

from emu_ops import EmuOps
from tooncrafter_amd import ops
from tooncrafter_amd.lvdm.attention import ContextCache
from tooncrafter_amd.lvdm.common import Act
from test_ln_fusion_cpu import _make_block


def _run(ff_fused_c, ln_fusion_k=None):
    spatial, temporal = _make_block(96), _make_block(None)
    emu = EmuOps(round_bf16=True, ln_fusion_k=ln_fusion_k, ff_fused_c=ff_fused_c)
    prev = ops.set_backend(emu)
    try:
        b, t, h, w = 1, 4, 2, 3
        g = torch.Generator().manual_seed(1)
        x = torch.randn(b * t * h * w, 320, generator=g).to(torch.bfloat16)
        act = Act(x, b, t, h, w)
        ctx = ContextCache(torch.randn(b, 77 + 16 * t, 96, generator=g), t)
        with torch.no_grad():
            ys = spatial.forward_spatial(x, act, ctx).float()
            yt = temporal.forward_temporal(x, act).float()
    finally:
        ops.set_backend(prev)
    return ys, yt, emu


def test_feed_forward_takes_the_fused_operator_when_offered():
    ys0, yt0, e0 = _run(None)
    ys1, yt1, e1 = _run(320)
    assert e0.ff_fused_calls == 0 and e1.ff_fused_calls == 2            # one feed-forward per block, spatial + temporal
    assert e1.ln_fused_calls == 2                                       # ... whose LayerNorm went with it (the emulation
    rel = lambda a, b: float((a - b).norm() / b.norm())                 # counts it in its first GEMM), and no other
    print("fused feed-forward on/off: spatial", rel(ys1, ys0), "temporal", rel(yt1, yt0))
    assert rel(ys1, ys0) < 1e-2 and rel(yt1, yt0) < 1e-2                 # bf16 rounding of w * gamma vs of gamma * x_hat


def test_other_widths_keep_the_three_launches():
    _, _, e = _run(640)                                                 # offered for another width only: not taken
    assert e.ff_fused_calls == 0
    _, _, e = _run(320, ln_fusion_k=320)                                # with the GEMM-prologue LayerNorm on as well
    assert e.ff_fused_calls == 2 and e.ln_fused_calls == 6


def test_fused_mirror_equals_the_chain_it_replaces():
    """EmuOps.ff_geglu_fused against LayerNorm -> GEGLU projection -> ff2 + residual spelled out in fp64."""
    from tooncrafter_amd.lvdm.common import fold_layernorm, pack_geglu, pack_linear
    g = torch.Generator().manual_seed(3)
    c, hid, m = 320, 1280, 50
    x = (torch.randn(m, c, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w1, b1 = torch.randn(2 * hid, c, generator=g) * 0.05, torch.randn(2 * hid, generator=g) * 0.1
    w2, b2 = torch.randn(c, hid, generator=g) * 0.03, torch.randn(c, generator=g) * 0.1
    gamma, beta = torch.randn(c, generator=g) * 0.2 + 1.0, torch.randn(c, generator=g) * 0.1
    wf, bf = fold_layernorm(w1, b1, gamma, beta)
    pw, pb = pack_geglu(wf, bf)
    emu = EmuOps(round_bf16=True, ff_fused_c=320)
    got = emu.ff_geglu_fused(x, pw, pb, pack_linear(w2), b2.float(), ln_eps=1e-5).double()
    xd = x.double()
    h = torch.nn.functional.layer_norm(xd, (c,), gamma.double(), beta.double(), 1e-5) @ w1.double().t() + b1.double()
    ref = xd + (h[:, :hid] * torch.nn.functional.gelu(h[:, hid:])) @ w2.double().t() + b2.double()
    err = float((got - ref).norm() / ref.norm())
    print("fused mirror vs fp64 chain: rel-L2", err)
    assert err < 6e-3


def test_temporal_attention_takes_the_fused_operator_when_offered():
    """BasicTransformerBlock.forward_temporal on the emulation: both self-attentions over the frames go through
    temporal_attn_fused (ABI 9, csrc/tb_fused.hip) when the backend offers it for the width, with the LayerNorm-folded qkv."""
    from test_ln_fusion_cpu import _make_block
    temporal = _make_block(None)
    outs = {}
    for c in (None, 320, 640):
        emu = EmuOps(round_bf16=True, tb_fused_c=c)
        prev = ops.set_backend(emu)
        try:
            b, t, h, w = 1, 16, 2, 4
            x = torch.randn(b * t * h * w, 320, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)
            with torch.no_grad():
                outs[c] = temporal.forward_temporal(x, Act(x, b, t, h, w)).float()
        finally:
            ops.set_backend(prev)
        assert emu.tb_fused_calls == (2 if c == 320 else 0), (c, emu.tb_fused_calls)
    rel = float((outs[320] - outs[None]).norm() / outs[None].norm())
    print("fused temporal attention on/off (emulation):", rel)
    assert rel < 1e-2 and torch.equal(outs[640], outs[None])


def test_temporal_attention_takes_the_qkv_attention_launch_when_offered():
    """Where the level-0 fusion does not apply, CrossAttention.forward_temporal_self hands the projection AND the attentions
    to temporal_qkv_attn (ABI 13, csrc/qkv_attn.hip) when the backend offers it -- the same arithmetic as gemm +
    attention_temporal, so on the emulation the block's output is the same bits; a backend that also offers the level-0
    fusion for the width keeps that one (it swallows more)."""
    from test_ln_fusion_cpu import _make_block
    temporal = _make_block(None)
    outs, calls = {}, {}
    for key, kw in (("off", {}), ("tqa", dict(tqa=True)), ("both", dict(tqa=True, tb_fused_c=320))):
        emu = EmuOps(round_bf16=True, **kw)
        prev = ops.set_backend(emu)
        try:
            b, t, h, w = 1, 16, 2, 4
            x = torch.randn(b * t * h * w, 320, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)
            with torch.no_grad():
                outs[key] = temporal.forward_temporal(x, Act(x, b, t, h, w)).float()
        finally:
            ops.set_backend(prev)
        calls[key] = (emu.tqa_calls, emu.tb_fused_calls)
    assert calls == {"off": (0, 0), "tqa": (2, 0), "both": (0, 2)}, calls
    assert torch.equal(outs["tqa"], outs["off"])
