"""The temporal q / k / v projection and its 16 x 16 attentions as ONE launch (ABI 13: tc_temporal_qkv_attn,
csrc/qkv_attn.hip; reference lvdm/modules/attention.py:96-134 over the 16 frames of a pixel, called with context = None
from TemporalTransformer attention.py:365-412).

Checked against (a) the two launches it replaces -- tc_gemm_bf16 (fused qkv) + tc_attn_temporal; (b) the emulated operator;
(c) the fp64 statement of the reference's attention.  Shapes: the BASELINE geometries of UNet levels 1 / 2 / 3 (B = 2,
16 frames, 20 x 32 / 10 x 16 / 5 x 8 pixels at C = 640 / 1280 / 1280), level 0's width, one tile, a tile count that is not a
multiple of 8 (surplus blocks exit), rows at a wider pitch, with and without a bias; repeated launches are bit-identical;
both bindings give the same bits.
"""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd
from tooncrafter_amd.lvdm.common import pack_linear

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
T = 16


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


def _w(c, seed=1):
    raw = torch.cat([rnd(c, c, seed=seed + i, scale=1.4 * c ** -0.5, dtype=torch.float32) for i in range(3)], 0)
    return raw, pack_linear(raw)


def _x(b, hw, c, seed=11, pitch=None):
    full = rnd(b * T * hw, pitch or c, seed=seed, scale=1.2) + 0.1
    return full.to(BF16)[:, :c]


# tag, b, hw, c, pitch
CASES = [("level 1 (B = 2, 20 x 32, C = 640)", 2, 640, 640, None), ("level 2 (B = 2, 10 x 16, C = 1280)", 2, 160, 1280, None),
         ("level 3 (B = 2, 5 x 8, C = 1280)", 2, 40, 1280, None), ("level-0 width (C = 320), B = 1, 24 pixels", 1, 24, 320, None),
         ("one tile (8 pixels), C = 640", 1, 8, 640, None), ("tiles not a multiple of 8 (B = 3, 104 pixels), C = 640", 3, 104, 640, None),
         ("rows at pitch 1920, C = 640", 1, 64, 640, 1920), ("one head (C = 64)", 1, 16, 64, None)]


@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
@pytest.mark.parametrize("tag,b,hw,c,pitch", CASES, ids=[c[0] for c in CASES])
def test_fused_vs_two_launches_and_emulation(hip, tag, b, hw, c, pitch, bias):
    heads = c // 64
    x = _x(b, hw, c, pitch=pitch)
    raw, w = _w(c)
    bq = rnd(3 * c, seed=9, scale=0.2, dtype=torch.float32) if bias else None
    assert hip.temporal_qkv_attn_eligible(b=b, t=T, hw=hw, c=c, heads=heads, ldx=x.stride(0))
    kw = dict(b=b, t=T, hw=hw, heads=heads)
    out = hip.temporal_qkv_attn(x, w, bq, **kw)
    torch.cuda.synchronize()
    ref = hip.attention_temporal(hip.gemm(x.contiguous(), w, bq), **kw)
    d = (out.float() - ref.float()).abs()
    scale = float(ref.float().abs().max())
    print(f"{tag} / {'bias' if bias else 'no bias'}: fused vs two launches: max |d| {float(d.max()):.3e} "
          f"({float(d.max()) / (scale * 2 ** -8):.2f} bf16-ulp of scale), {int((d > 0).sum())} of {d.numel()} elements differ")
    check(out, ref, f"{tag}: fused vs gemm + attention_temporal", rel=3e-3)
    if b * T * hw * c <= 8192 * 640:
        emu = EmuOps(round_bf16=True, tqa=True)
        want = emu.temporal_qkv_attn(x.cpu(), w.cpu(), None if bq is None else bq.cpu(), **kw)
        check(out.cpu(), want, f"{tag}: fused vs emulation")
    again = hip.temporal_qkv_attn(x, w, bq, **kw)
    assert torch.equal(out, again), "repeated launch differs"


def test_fused_vs_fp64_reference_attention(hip):
    """softmax(q k^T / 8) v over the frames of every pixel, q / k / v = Linear(x), heads re-concatenated, in fp64."""
    b, hw, c = 1, 48, 640
    heads = c // 64
    x = _x(b, hw, c, seed=21)
    raw, w = _w(c, seed=5)
    out = hip.temporal_qkv_attn(x, w, None, b=b, t=T, hw=hw, heads=heads).double().cpu()
    qkv = x.double().cpu() @ raw.double().cpu().t()
    q, k, v = (qkv[:, i * c:(i + 1) * c].reshape(b, T, hw, heads, 64).permute(0, 2, 3, 1, 4) for i in range(3))
    o = ((q @ k.transpose(-1, -2)) * 64 ** -0.5).softmax(-1) @ v                       # [b, hw, heads, T, 64]
    ref = o.permute(0, 3, 1, 2, 4).reshape(b * T * hw, c)
    err = float((out - ref).norm() / ref.norm())
    print(f"fused qkv + temporal attention vs fp64 reference: rel-L2 {err:.3e}")
    assert err < 6e-3


def test_attention_is_over_frames_of_the_same_pixel(hip):
    """Changing ONE row changes the 16 output rows of its pixel and nothing else (gather / tile-row mapping / head slices)."""
    b, hw, c = 2, 40, 640
    x = _x(b, hw, c, seed=31).contiguous()
    _, w = _w(c, seed=7)
    kw = dict(b=b, t=T, hw=hw, heads=c // 64)
    y0 = hip.temporal_qkv_attn(x, w, None, **kw)
    x2 = x.clone()
    pix, bb = 13, 1
    rows = (bb * T + torch.arange(T, device=x.device)) * hw + pix
    x2[rows[5]] = (x2[rows[5]].float() * -0.7 + 0.2).to(BF16)                         # frame 5 of pixel 13 of clip 1
    y1 = hip.temporal_qkv_attn(x2, w, None, **kw)
    changed = (y0 != y1).any(dim=1).nonzero().flatten().tolist()
    assert set(changed) <= set(rows.tolist()) and len(changed) >= T - 1, changed
    cols = (y0[rows] != y1[rows]).any(dim=0)
    assert int(cols.sum()) > c // 2                                                     # every head's slice moved


def test_eligibility_and_refusals(hip, monkeypatch):
    from tooncrafter_amd._lib import TooncrafterHipError
    assert hip.temporal_qkv_attn_eligible(b=2, t=16, hw=640, c=640, heads=10)
    assert not hip.temporal_qkv_attn_eligible(b=2, t=8, hw=640, c=640, heads=10)           # 16 frames only
    assert not hip.temporal_qkv_attn_eligible(b=2, t=16, hw=636, c=640, heads=10)          # hw % 8
    assert not hip.temporal_qkv_attn_eligible(b=2, t=16, hw=640, c=640, heads=8)           # c = heads * 64
    x = _x(1, 8, 640)
    _, w = _w(640)
    with pytest.raises(ValueError):
        hip.temporal_qkv_attn(x, w[:640], None, b=1, t=T, hw=8, heads=10)
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.temporal_qkv_attn(x.cpu(), w, None, b=1, t=T, hw=8, heads=10)
    with pytest.raises(TooncrafterHipError):
        hip.temporal_qkv_attn(_x(1, 4, 640), w, None, b=1, t=T, hw=4, heads=10)            # hw % 8: TC_ESHAPE, nothing launched
    monkeypatch.setenv("TC_QKV_ATTN", "0")
    assert not hip.temporal_qkv_attn_eligible(b=2, t=16, hw=640, c=640, heads=10)


def test_custom_op_binding_gives_the_same_bits(hip):
    from tooncrafter_amd import torch_ops
    t = torch_ops.TorchLibOps()
    b, hw, c = 1, 64, 640
    x, (_, w) = _x(b, hw, c, seed=41), _w(c, seed=3)
    bq = rnd(3 * c, seed=9, scale=0.2, dtype=torch.float32)
    kw = dict(b=b, t=T, hw=hw, heads=c // 64)
    assert torch.equal(t.temporal_qkv_attn(x, w, None, **kw), hip.temporal_qkv_attn(x, w, None, **kw))
    assert torch.equal(t.temporal_qkv_attn(x, w, bq, **kw), hip.temporal_qkv_attn(x, w, bq, **kw))


def test_block_routes_through_the_fused_operator(hip, monkeypatch):
    """A level-1 BasicTransformerBlock (temporal flavour) on the HIP backend, qkv + attention as one launch on vs off."""
    from tooncrafter_amd import ops
    from tooncrafter_amd.lvdm.attention import BasicTransformerBlock
    from tooncrafter_amd.lvdm.common import Act
    torch.manual_seed(0)
    blk = BasicTransformerBlock(640, 10, 64, context_dim=None).eval()
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
        calls = []
        real = hip.temporal_qkv_attn
        monkeypatch.setattr(hip, "temporal_qkv_attn", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            y1 = blk.forward_temporal(x, act)
            monkeypatch.setenv("TC_QKV_ATTN", "0")
            y0 = blk.forward_temporal(x, act)
        assert len(calls) == 2
        check(y1, y0, "temporal block, qkv + attention as one launch on vs off", rel=8e-3)
    finally:
        ops.set_backend(prev)
