"""oracle/mx.py (MXFP8 operand format of the configs[4] GEMM path) against hand-computed known answers of the
OCP MX v1.0 / OCP FP8 encodings, plus the properties the GPU tests rely on.  No GPU."""
import torch

from oracle import mx


def q1(vals):
    x = torch.zeros(1, 32)
    x[0, :len(vals)] = torch.tensor(vals)
    q, s = mx.quantize_mxfp8(x)
    return q[0, :len(vals)].tolist(), int(s[0, 0])


def test_known_answers():
    # amax = 448 = 1.75 * 2^8: shared exponent 8 - 8 = 0 -> scale byte 127, elements stored as they are
    q, s = q1([448.0, 1.0, -1.0, 0.5, 2.0 ** -9, 0.0])
    assert s == 127 and q == [0x7E, 0x38, 0xB8, 0x30, 0x01, 0x00]
    # amax = 1.0: shared exponent -8 -> byte 119, 1.0 is stored as 256 = 2^8 -> exponent field 15, mantissa 0
    q, s = q1([1.0, 0.75, -0.5])
    assert s == 119 and q == [0x78, 0x74, 0xF0]
    # amax = 1.9375 * 2^8 = 496 (same binade as 448): scale 1, 496 saturates to 448 (0x7E), not NaN (0x7F)
    q, s = q1([496.0, 480.0, 464.0])
    assert s == 127 and q[0] == 0x7E and q[1] == 0x7E
    # round to nearest even at 3 mantissa bits: 1 + 1/16 is a tie between 1.0 (even mantissa) and 1.125
    q, s = q1([256.0, 256.0 * (1 + 1 / 16), 256.0 * (1 + 3 / 16)])
    assert s == 127 and q == [0x78, 0x78, 0x7A]
    # an all-zero block: smallest scale, zero elements
    q, s = q1([0.0, 0.0])
    assert s == 0 and q == [0, 0]


def test_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g) * torch.exp(3.0 * torch.randn(64, 1, generator=g))
    q, s = mx.quantize_mxfp8(x)
    d = mx.dequantize_mxfp8(q, s)
    # per block: the largest element keeps >= 3 mantissa bits: |err| <= 2^-4 amax (elements in 448..512 saturate: 12.5 %)
    xb, db = x.reshape(64, 8, 32), d.reshape(64, 8, 32)
    assert bool(((xb - db).abs().amax(2) <= 0.125 * xb.abs().amax(2)).all())
    assert float((d - x).norm() / x.norm()) < 0.04
    # idempotent, and exactly representable in bf16 (what tests/test_gpu_fp8.py's reference relies on)
    q2, s2 = mx.quantize_mxfp8(d)
    assert torch.equal(q, q2) and torch.equal(s, s2)
    assert torch.equal(d.to(torch.bfloat16).float(), d)
    # a GEMM through the format: ~2^-4.6 relative noise per operand
    a, w = torch.randn(128, 512, generator=g), torch.randn(64, 512, generator=g)
    rel = float((mx.gemm_mx(a, w) - a @ w.t()).norm() / (a @ w.t()).norm())
    assert 0.02 < rel < 0.06
