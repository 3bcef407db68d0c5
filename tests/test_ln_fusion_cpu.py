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
