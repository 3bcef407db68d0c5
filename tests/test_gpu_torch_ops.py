"""`torch.ops.tooncrafter.*` (TORCH_LIBRARY layer, csrc/torch_ops.cpp) against the ctypes binding of the same C ABI:
the same kernels behind both, so results must be bit-identical; plus a tiny UNet forward with the torch-op backend."""
import os

import pytest
import torch

from conftest import TINY_UNET_CFG, sub_state_dict
from tooncrafter_amd import ops, synth
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU

pytestmark = pytest.mark.gpu
DEV, BF16 = "cuda", torch.bfloat16


@pytest.fixture(scope="module")
def both():
    from tooncrafter_amd.ops import HipOps
    from tooncrafter_amd.torch_ops import TorchLibOps
    return HipOps(), TorchLibOps()


def rnd(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def test_torch_ops_match_ctypes_binding_bitwise(both):
    c, t = both
    a, w, b = rnd(640, 320, seed=1), rnd(960, 320, seed=2, scale=0.05), rnd(960, seed=3, dtype=torch.float32)
    res, rb = rnd(640, 960, seed=4), rnd(5, 960, seed=5, dtype=torch.float32)
    kw = dict(act=ACT_SILU, residual=res, row_bias=rb, row_div=128, alpha=0.5, out_scale=0.75)
    assert torch.equal(t.gemm(a, w, b, **kw), c.gemm(a, w, b, **kw))
    from tooncrafter_amd.lvdm.common import pack_geglu
    wp, bp = pack_geglu(torch.randn(2560, 320) * 0.05, torch.randn(2560))
    wp, bp = wp.to(DEV), bp.to(DEV)
    assert torch.equal(t.gemm(a, wp, bp, act=ACT_GEGLU), c.gemm(a, wp, bp, act=ACT_GEGLU))
    x = rnd(3 * 5 * 8, 128, seed=6)
    wc = rnd(320, 9 * 128, seed=7, scale=0.03)
    geom = dict(kind="3x3", frames=3, cin=128, h_in=5, w_in=8, h_out=5, w_out=8, stride=1, upsample=False)
    assert torch.equal(t.gemm(x, wc, None, conv=geom, out_f32=True), c.gemm(x, wc, None, conv=geom, out_f32=True))
    gt = dict(kind="t3", frames=6, t_len=3, cin=128, h_out=1, w_out=20)
    wt = rnd(128, 3 * 128, seed=8, scale=0.05)
    assert torch.equal(t.gemm(x, wt, None, conv=gt), c.gemm(x, wt, None, conv=gt))
    q, kv, kv2 = rnd(8 * 100, 128, seed=9), rnd(2 * 77, 256, seed=10), rnd(8 * 16, 256, seed=11)
    kwa = dict(batch=8, heads=2, lq=100, lk=77, kv_bdiv=4)
    assert torch.equal(t.attention(q, kv[:, :128], kv[:, 128:], **kwa), c.attention(q, kv[:, :128], kv[:, 128:], **kwa))
    kw2 = dict(k2=kv2[:, :128], v2=kv2[:, 128:], lk2=16, kv2_bdiv=1)
    assert torch.equal(t.attention(q, kv[:, :128], kv[:, 128:], **kwa, **kw2), c.attention(q, kv[:, :128], kv[:, 128:], **kwa, **kw2))
    qkv = rnd(2 * 4 * 30, 3 * 128, seed=12)
    assert torch.equal(t.attention_temporal(qkv, b=2, t=4, hw=30, heads=2), c.attention_temporal(qkv, b=2, t=4, hw=30, heads=2))
    xs = rnd(4 * 160, 1280, seed=13)
    g, be = rnd(1280, seed=14, dtype=torch.float32), rnd(1280, seed=15, dtype=torch.float32)
    assert torch.equal(t.groupnorm(xs, g, be, samples=4, rows=160, eps=1e-5, silu=True),
                       c.groupnorm(xs, g, be, samples=4, rows=160, eps=1e-5, silu=True))
    assert torch.equal(t.layernorm(xs, g, be), c.layernorm(xs, g, be))
    # ABI 12: the consumer's weights as a prefetch list -- same policy, same result through both bindings
    wpf = [rnd(1280, 2304, seed=40), rnd(640, 1280, seed=41)]
    assert [x.data_ptr() for x in t.prefetch_list(xs.shape[0], wpf)] == [x.data_ptr() for x in wpf] and not t.prefetch_list(81920, wpf)
    assert torch.equal(t.groupnorm(xs, g, be, samples=4, rows=160, eps=1e-5, silu=True, prefetch=wpf),
                       c.groupnorm(xs, g, be, samples=4, rows=160, eps=1e-5, silu=True))
    assert torch.equal(t.layernorm(xs, g, be, prefetch=wpf), c.layernorm(xs, g, be, prefetch=wpf))
    lat = [rnd(2, 4, 4, 8, 8, seed=s, dtype=torch.float32) for s in (16, 17, 18, 19)]
    sc = dict(cfg_scale=7.5, guidance_rescale=0.7, sqrt_ac=0.6, sqrt_1m_ac=0.8, sqrt_a_prev=0.7, dir_coef=0.5, sigma=0.3, x0_rescale=0.98)
    for u, v in zip(t.ddim_step(*lat, **sc), c.ddim_step(*lat, **sc)):
        assert torch.equal(u, v)
    # MXFP8 pair through the op layer: same bytes as the ctypes binding, same GEMM result as HipOps' fp8 route
    aq_t, as_t = t.quant_mxfp8(a, 320)
    aq_c, as_c = c.quant_mxfp8(a, 320)
    assert torch.equal(aq_t, aq_c) and torch.equal(as_t, as_c)
    wq, ws = t.quant_mxfp8(w, 320)
    y_t = t.t.gemm_mx(aq_t, as_t, wq, ws, b, res, None, 0, ACT_NONE, 1.0, 1.0, False, [])
    c.fp8, c.fp8_min_k, c.fp8_min_n, c.fp8_min_m, c.fp8_n_over_k = "linear", 0, 0, 1, 0.0
    try:
        assert torch.equal(y_t, c.gemm(a, w, b, residual=res))
    finally:
        c.fp8 = None
    with pytest.raises(RuntimeError):
        t.gemm(a[:, :60], w[:, :60].contiguous())                  # K not a multiple of 8: the C ABI's TC_EALIGN surfaces
    # the level-0 one-launch operators (ABI 9) through the op layer: same launches, same bits
    x0 = rnd(2 * 16 * 40, 320, seed=20)
    w1, b1 = pack_geglu(torch.randn(2560, 320) * 0.05, torch.randn(2560))
    w1, b1 = w1.to(DEV), b1.to(DEV)
    w2, b2 = rnd(320, 1280, seed=21, scale=0.03), rnd(320, seed=22, dtype=torch.float32)
    assert torch.equal(t.ff_geglu_fused(x0, w1, b1, w2, b2, ln_eps=1e-5), c.ff_geglu_fused(x0, w1, b1, w2, b2, ln_eps=1e-5))
    wqkv, bqkv = rnd(960, 320, seed=23, scale=0.05), rnd(960, seed=24, dtype=torch.float32)
    wo, bo = rnd(320, 320, seed=25, scale=0.05), rnd(320, seed=26, dtype=torch.float32)
    kwt = dict(b=2, t=16, hw=40, heads=5, ln_eps=1e-5)
    os.environ["TC_TB_FUSED"] = "1"                  # the level-0 single launch is opt-in since round 6 (the library reads the switch per call)
    try:
        assert torch.equal(t.temporal_attn_fused(x0, wqkv, bqkv, wo, bo, **kwt), c.temporal_attn_fused(x0, wqkv, bqkv, wo, bo, **kwt))
    finally:
        os.environ.pop("TC_TB_FUSED", None)
    # ABI 13: the default route of every temporal self-attention
    kwq = dict(b=2, t=16, hw=40, heads=5)
    assert torch.equal(t.temporal_qkv_attn(x0, wqkv, None, **kwq), c.temporal_qkv_attn(x0, wqkv, None, **kwq))
    assert torch.equal(t.temporal_qkv_attn(x0, wqkv, bqkv, **kwq), c.temporal_qkv_attn(x0, wqkv, bqkv, **kwq))


def test_tiny_unet_through_torch_op_layer(both, tiny_sd):
    from tooncrafter_amd.lvdm.openaimodel3d import UNetModel
    c, t = both
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    un = un.to(DEV)
    inp = synth.synth_inputs(2, 4, 8, 8, context_dim=96, seed=3)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1).to(DEV)
    ts = torch.tensor([601, 601], device=DEV)
    ctx, fs = inp["cond"].to(DEV), inp["fs"].to(DEV)
    with torch.no_grad():
        prev = ops.set_backend(c)
        try:
            y_c = un(x, ts, context=ctx, fs=fs).clone()
            ops.set_backend(t)
            un._ctx_cache = None
            y_t = un(x, ts, context=ctx, fs=fs).clone()
        finally:
            ops.set_backend(prev)
    assert torch.isfinite(y_t).all() and torch.equal(y_c, y_t)
