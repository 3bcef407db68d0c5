"""LayerNorm folded into its consumer GEMM (ABI 8: TcGemmParams.a_norm; reference lvdm/modules/attention.py:225-227,
242-246): the host half -- common.fold_layernorm and BasicTransformerBlock._pre -- on the CPU emulation."""
import torch

from emu_ops import EmuOps
from tooncrafter_amd import ops
from tooncrafter_amd.lvdm.attention import BasicTransformerBlock, ContextCache
from tooncrafter_amd.lvdm.common import Act, fold_layernorm


def test_fold_layernorm_is_exact_in_real_arithmetic():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 320, generator=g, dtype=torch.float64) * 2 + 0.7
    w = torch.randn(96, 320, generator=g, dtype=torch.float64)
    b = torch.randn(96, generator=g, dtype=torch.float64)
    gamma, beta = torch.randn(320, generator=g, dtype=torch.float64), torch.randn(320, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.layer_norm(x, (320,), gamma, beta, 1e-5) @ w.t() + b
    wf, bf = fold_layernorm(w.float(), b.float(), gamma.float(), beta.float())
    xn = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    got = xn @ wf.double().t() + bf.double()
    assert float((got - ref).abs().max()) < 1e-4
    wf2, bf2 = fold_layernorm(w.float(), None, gamma.float(), beta.float())
    assert torch.allclose(bf2.double(), w @ beta, atol=1e-4)


def _make_block(context_dim):
    torch.manual_seed(0)
    blk = BasicTransformerBlock(320, 5, 64, context_dim=context_dim, image_cross_attention=context_dim is not None).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.05)
        for i in (1, 2, 3):
            getattr(blk, f"norm{i}").weight.add_(1.0)
    return blk


def _block_outputs(fuse):
    spatial, temporal = _make_block(96), _make_block(None)       # SpatialTransformer / TemporalTransformer flavours
    emu = EmuOps(round_bf16=True, ln_fusion_k=320 if fuse else None)
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
    return ys, yt, emu.ln_fused_calls


def test_block_with_and_without_ln_fusion_agree():
    ys0, yt0, n0 = _block_outputs(False)
    ys1, yt1, n1 = _block_outputs(True)
    assert n0 == 0 and n1 == 6                      # three LayerNorms per pass, spatial + temporal
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print("LN fusion on/off: spatial", rel(ys1, ys0), "temporal", rel(yt1, yt0))
    assert rel(ys1, ys0) < 1e-2 and rel(yt1, yt0) < 1e-2      # bf16 rounding of w * gamma vs of gamma * x_hat


def test_folded_weights_follow_a_reload_of_the_child_or_of_the_norm():
    """the folded consumer weights are built from the CHILD's parameters (attn.to_q/k/v, ff.net[0].proj) and the
    block's norm; reloading either alone must rebuild them (the block-level cache kept using the old ones)."""
    blk = _make_block(None)
    emu = EmuOps(round_bf16=True, ln_fusion_k=320)
    prev = ops.set_backend(emu)
    try:
        b, t, h, w = 1, 4, 2, 3
        x = torch.randn(b * t * h * w, 320, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
        act = Act(x, b, t, h, w)
        with torch.no_grad():
            y0 = blk.forward_temporal(x, act).float()
            sd = {k: v.clone() * 0.5 for k, v in blk.ff.state_dict().items()}
            blk.ff.load_state_dict(sd)                          # the child alone
            y1 = blk.forward_temporal(x, act).float()
            fresh = _make_block(None)
            fresh.ff.load_state_dict(sd)
            y1_ref = fresh.forward_temporal(x, act).float()
            blk.norm3.load_state_dict({"weight": blk.norm3.weight * 2.0, "bias": blk.norm3.bias + 0.3})   # the norm alone
            fresh.norm3.load_state_dict(blk.norm3.state_dict())
            y2, y2_ref = blk.forward_temporal(x, act).float(), fresh.forward_temporal(x, act).float()
    finally:
        ops.set_backend(prev)
    assert not torch.equal(y0, y1)
    assert torch.equal(y1, y1_ref), "stale folded GEGLU weights after reloading ff alone"
    assert torch.equal(y2, y2_ref), "stale folded weights after reloading the norm alone"


def test_context_cache_is_stale_after_new_weights():
    """cached K/V were projected with the weights of their epoch; load_state_dict bumps the epoch and the cache
    must refresh although the conditioning tensor object is unchanged."""
    from tooncrafter_amd.lvdm.common import PackedModule
    ctx_t = torch.randn(1, 77 + 16 * 4, 96, generator=torch.Generator().manual_seed(2))
    cache = ContextCache(ctx_t, 4)
    assert cache.is_current(ctx_t)
    blk = _make_block(96)
    blk.load_state_dict(blk.state_dict())                       # any reload: the epoch moves on
    assert cache.epoch != PackedModule.graph_epoch()
    assert not cache.is_current(ctx_t)
    cache.refresh(ctx_t)
    assert cache.is_current(ctx_t)
