"""The text + image cross-attention with its QUERY PROJECTION inside the launch (ABI 13: tc_attn_d64_qproj, csrc/attention.hip
attn_d64_dma_kernel<DUAL, QP = true>; reference lvdm/modules/attention.py:96 `q = self.to_q(x)` in front of the two softmaxes
of attention.py:153-207).

Checked against (a) the two launches it replaces -- tc_gemm_bf16 (to_q) + tc_attn_d64 -- which it must equal BIT FOR BIT (the
projected tile is rounded to bf16 exactly as the GEMM's store rounds it, sums in the same K order, and the attention body is
the same code); (b) the emulated operator.  Shapes: the BASELINE geometries of UNet levels 1 / 2 / 3 (32 frames, 640 / 160 / 40
queries per frame, 77 text keys shared by the 16 frames of a clip + 16 image keys per frame), level 0's width, a ragged last
query tile, a single key set, rows at a wider pitch; repeated launches are bit-identical; both bindings give the same bits.
"""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd
from tooncrafter_amd.lvdm.common import pack_linear

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


def _case(batch, lq, c, t=16, lk=77, lk2=16, pitch=None, seed=0):
    heads = c // 64
    x = (rnd(batch * lq, pitch or c, seed=seed + 1, scale=1.1) + 0.05).to(BF16)[:, :c]
    wq = pack_linear(rnd(c, c, seed=seed + 2, scale=1.3 * c ** -0.5, dtype=torch.float32))
    kvb = (batch + t - 1) // t
    kv = rnd(kvb * lk, 2 * c, seed=seed + 3)
    kv2 = rnd(batch * lk2, 2 * c, seed=seed + 4) if lk2 else None
    kw = dict(batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=t)
    if lk2:
        kw.update(k2=kv2[:, :c], v2=kv2[:, c:], lk2=lk2, kv2_bdiv=1)
    return x, wq, kv[:, :c], kv[:, c:], kw


# tag, batch (frames), lq, c, lk2, pitch
CASES = [("level 1 (32 frames x 640 queries, C = 640)", 32, 640, 640, 16, None), ("level 2 (32 x 160, C = 1280)", 32, 160, 1280, 16, None),
         ("level 3 (32 x 40, C = 1280)", 32, 40, 1280, 16, None), ("level-0 width (4 x 300 queries, C = 320): ragged last tile", 4, 300, 320, 16, None),
         ("text keys only, C = 640", 16, 130, 640, 0, None), ("rows at pitch 1920, C = 640", 3, 64, 640, 16, 1920),
         ("one head (C = 64)", 2, 129, 64, 16, None)]


@pytest.mark.parametrize("tag,batch,lq,c,lk2,pitch", CASES, ids=[c[0] for c in CASES])
def test_qproj_vs_two_launches_and_emulation(hip, tag, batch, lq, c, lk2, pitch):
    x, wq, k, v, kw = _case(batch, lq, c, lk2=lk2, pitch=pitch)
    assert hip.attention_qproj_eligible(x, wq, k, v, **kw)
    out = hip.attention_qproj(x, wq, k, v, **kw)
    torch.cuda.synchronize()
    ref = hip.attention(hip.gemm(x.contiguous(), wq), k, v, **kw)
    d = (out.float() - ref.float()).abs()
    print(f"{tag}: one launch vs gemm + attention: max |d| {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} elements differ")
    assert torch.equal(out, ref), "the fused launch must reproduce gemm + attention bit for bit"
    if batch * lq * c <= 4096 * 640:
        emu = EmuOps(round_bf16=True, qproj=True)
        cpu = lambda t: None if t is None else t.cpu()
        kwc = {kk: (cpu(vv) if torch.is_tensor(vv) else vv) for kk, vv in kw.items()}
        check(out.cpu(), emu.attention_qproj(x.cpu(), wq.cpu(), k.cpu(), v.cpu(), **kwc), f"{tag}: one launch vs emulation", rel=8e-3)
    assert torch.equal(out, hip.attention_qproj(x, wq, k, v, **kw)), "repeated launch differs"


def test_eligibility_and_refusals(hip, monkeypatch):
    from tooncrafter_amd._lib import TooncrafterHipError
    x, wq, k, v, kw = _case(2, 64, 640)
    assert hip.attention_qproj_eligible(x, wq, k, v, **kw)
    with pytest.raises(ValueError):
        hip.attention_qproj(x, wq[:320], k, v, **kw)                                   # wq must be [heads*64, c]
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.attention_qproj(x.cpu(), wq, k, v, **kw)
    monkeypatch.setenv("TC_ATTN_QPROJ", "0")
    assert not hip.attention_qproj_eligible(x, wq, k, v, **kw)
    with pytest.raises(TooncrafterHipError):
        hip.attention_qproj(x, wq, k, v, **kw)                                         # TC_ESHAPE: nothing launched


def test_custom_op_binding_gives_the_same_bits(hip):
    from tooncrafter_amd import torch_ops
    t = torch_ops.TorchLibOps()
    x, wq, k, v, kw = _case(5, 200, 640, seed=7)
    assert torch.equal(t.attention_qproj(x, wq, k, v, **kw), hip.attention_qproj(x, wq, k, v, **kw))


def test_block_routes_through_the_fused_operator(hip, monkeypatch):
    """A level-1 BasicTransformerBlock (spatial flavour, text + image context) on the HIP backend: to_q inside the attention on vs off."""
    from tooncrafter_amd import ops
    from tooncrafter_amd.lvdm.attention import BasicTransformerBlock, ContextCache
    from tooncrafter_amd.lvdm.common import Act
    torch.manual_seed(0)
    blk = BasicTransformerBlock(640, 10, 64, context_dim=1024, video_length=16, image_cross_attention=True).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.04)
        for i in (1, 2, 3):
            getattr(blk, f"norm{i}").weight.add_(1.0)
    blk = blk.cuda()
    prev = ops.set_backend(hip)
    try:
        b, t, h, w = 1, 16, 8, 16
        x = rnd(b * t * h * w, 640, seed=41)
        act = Act(x, b, t, h, w)
        context = rnd(b, 77 + 16 * t, 1024, seed=42).float()            # 77 text tokens + 16 image tokens per frame
        calls = []
        real = hip.attention_qproj
        monkeypatch.setattr(hip, "attention_qproj", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            y1 = blk.forward_spatial(x, act, ContextCache(context, t))
            monkeypatch.setenv("TC_ATTN_QPROJ", "0")
            y0 = blk.forward_spatial(x, act, ContextCache(context, t))
        assert len(calls) == 1
        assert torch.equal(y1, y0), "to_q inside the attention must not change a bit of the block's output"
    finally:
        ops.set_backend(prev)
