"""Size-independent properties of the operators at the sizes of BASELINE.json's configuration (B=2 UNet
forward of a 320x512x16f clip; the CPU oracle cannot reach these sizes in test time): linearity of the
implicit-GEMM convolutions, translation equivariance, the invariances of GroupNorm / LayerNorm, key-permutation
invariance of attention, bit-reproducibility.  Each property is exact in real arithmetic; the bounds are the
bf16 storage rounding of the tensors compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


def rnd(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def conv_geom(frames, h, w, cin):
    return dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)


def test_conv3x3_linearity_full_size(hip):
    """conv(x1 + x2) = conv(x1) + conv(x2) (no bias) on the level-0 layer: 32 frames of 40x64, 320 -> 320."""
    f, h, w, c = 32, 40, 64, 320
    x1, x2 = rnd(f * h * w, c, seed=1), rnd(f * h * w, c, seed=2)
    wt = rnd(c, 9 * c, seed=3, scale=(9 * c) ** -0.5)
    xs = (x1.float() + x2.float()).to(BF16)                          # rounded once; the property is about conv
    lhs = hip.gemm(xs, wt, conv=conv_geom(f, h, w, c), out_f32=True)
    rhs = hip.gemm(x1, wt, conv=conv_geom(f, h, w, c), out_f32=True) + hip.gemm(x2, wt, conv=conv_geom(f, h, w, c), out_f32=True)
    e = rel(lhs, rhs)
    print(f"conv3x3 linearity at 81920x320x2880: rel-L2 {e:.3e}")
    assert e < 4e-3                                                  # the bf16 rounding of x1 + x2


def test_conv3x3_translation_equivariance_full_size(hip):
    """Shifting every frame one pixel right shifts the output one pixel right (away from the borders)."""
    f, h, w, c = 32, 40, 64, 320
    x = rnd(f, h, w, c, seed=4)
    xs = torch.zeros_like(x)
    xs[:, :, 1:] = x[:, :, :-1]
    wt, b = rnd(c, 9 * c, seed=5, scale=(9 * c) ** -0.5), rnd(c, seed=6, dtype=torch.float32)
    y = hip.gemm(x.reshape(-1, c), wt, b, conv=conv_geom(f, h, w, c)).reshape(f, h, w, c)
    ys = hip.gemm(xs.reshape(-1, c), wt, b, conv=conv_geom(f, h, w, c)).reshape(f, h, w, c)
    assert torch.equal(ys[:, :, 2:-1], y[:, :, 1:-2])                # same products, same order: bit-identical


def test_temporal_conv_frame_shift_full_size(hip):
    """(3,1,1) convolution over T: delaying the clip by one frame delays the output by one frame."""
    b, t, hw, c = 2, 16, 2560, 320
    x = rnd(b, t, hw, c, seed=7)
    xs = torch.zeros_like(x)
    xs[:, 1:] = x[:, :-1]
    wt = rnd(c, 3 * c, seed=8, scale=(3 * c) ** -0.5)
    geom = dict(kind="t3", frames=b * t, t_len=t, cin=c, h_out=40, w_out=64)
    y = hip.gemm(x.reshape(-1, c), wt, conv=geom).reshape(b, t, hw, c)
    ys = hip.gemm(xs.reshape(-1, c), wt, conv=geom).reshape(b, t, hw, c)
    assert torch.equal(ys[:, 2:-1], y[:, 1:-2])


def test_groupnorm_affine_invariance_full_size(hip):
    """GroupNorm(a x + b) = GroupNorm(x) for a > 0 (clip-wide statistics, 2 x 40960 rows x 320 channels)."""
    s, rows, c = 2, 40960, 320
    x = rnd(s * rows, c, seed=9)
    g, be = rnd(c, seed=10, dtype=torch.float32) * 0.1 + 1.0, rnd(c, seed=11, dtype=torch.float32) * 0.1
    y = hip.groupnorm(x, g, be, samples=s, rows=rows, eps=1e-5, silu=True)
    x2 = (x.float() * 4.0 + 8.0).to(BF16)                            # exact in bf16 up to the shared exponent shift
    y2 = hip.groupnorm(x2, g, be, samples=s, rows=rows, eps=1e-5, silu=True)
    e = rel(y2, y)
    print(f"groupnorm affine invariance: rel-L2 {e:.3e}")
    assert e < 1e-2                                                  # 4x + 8 loses up to 2 mantissa bits of x


def test_layernorm_shift_invariance_and_rowwise_full_size(hip):
    rows, c = 81920, 320
    x = rnd(rows, c, seed=12)
    g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    y = hip.layernorm(x, g, be)
    # every row has zero mean / unit variance (gamma = 1, beta = 0)
    yf = y.float()
    assert float(yf.mean(1).abs().max()) < 2e-2 and float((yf.var(1, unbiased=False) - 1).abs().max()) < 3e-2
    # rows are independent: a permutation of the rows permutes the output
    perm = torch.randperm(rows, device=DEV, generator=torch.Generator(device=DEV).manual_seed(13))
    assert torch.equal(hip.layernorm(x[perm].contiguous(), g, be), y[perm])


def test_attention_key_permutation_invariance_full_size(hip):
    """softmax(q k^T) v does not depend on the order of the keys (level-0 self-attention: 2560 x 2560, 5 heads)."""
    b, heads, l = 4, 5, 2560
    q, k, v = (rnd(b * l, heads * 64, seed=s) for s in (14, 15, 16))
    o = hip.attention(q, k, v, batch=b, heads=heads, lq=l, lk=l)
    perm = torch.randperm(l, device=DEV, generator=torch.Generator(device=DEV).manual_seed(17))
    kp = k.reshape(b, l, -1)[:, perm].reshape(b * l, -1).contiguous()
    vp = v.reshape(b, l, -1)[:, perm].reshape(b * l, -1).contiguous()
    op = hip.attention(q, kp, vp, batch=b, heads=heads, lq=l, lk=l)
    e = rel(op, o)
    print(f"attention key-permutation invariance: rel-L2 {e:.3e}")
    assert e < 6e-3                                                  # different summation order + bf16 P rounding


def test_everything_is_bit_reproducible_full_size(hip):
    f, h, w, c = 32, 40, 64, 320
    x = rnd(f * h * w, c, seed=18)
    wt, b = rnd(c, 9 * c, seed=19, scale=(9 * c) ** -0.5), rnd(c, seed=20, dtype=torch.float32)
    g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    runs = []
    for _ in range(3):
        y = hip.gemm(x, wt, b, conv=conv_geom(f, h, w, c))
        y = hip.groupnorm(y, g, be, samples=f, rows=h * w, eps=1e-5, silu=True)
        y = hip.layernorm(y, g, be)
        y = hip.attention(y, y, y, batch=f, heads=5, lq=h * w, lk=h * w)
        runs.append(y)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
