"""The feed-forward of a level-0 transformer block as ONE launch (ABI 9: tc_ff_geglu_fused, csrc/ff_fused.hip; reference
lvdm/modules/attention.py:415-442 FeedForward / GEGLU behind norm3, attention.py:244-246).

Checked against (a) the three launches it replaces -- tc_layernorm, tc_gemm_bf16(GEGLU), tc_gemm_bf16(+residual): same
roundings, same order of the fp32 sums, so the two agree to a bf16 ulp or two of a few elements; (b) the emulated
operator (fp32 arithmetic, bf16 roundings where the kernels round); (c) the fp64 statement of the reference's block.
Shapes: the BASELINE level-0 row count (81920), a partial last tile, fewer rows than one tile, rows at a wider pitch,
with and without the LayerNorm; repeated launches are bit-identical (persistent blocks, no atomics).
"""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_ops import check, rnd
from tooncrafter_amd._lib import ACT_GEGLU
from tooncrafter_amd.lvdm.common import fold_layernorm, pack_geglu, pack_linear

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
C, HID = 320, 1280


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def weights():
    w1, b1 = rnd(2 * HID, C, seed=1, scale=0.05, dtype=torch.float32), rnd(2 * HID, seed=2, scale=0.1, dtype=torch.float32)
    w2, b2 = rnd(C, HID, seed=3, scale=0.03, dtype=torch.float32), rnd(C, seed=4, scale=0.1, dtype=torch.float32)
    gamma = rnd(C, seed=5, scale=0.2, dtype=torch.float32) + 1.0
    beta = rnd(C, seed=6, scale=0.1, dtype=torch.float32)
    wf, bf = fold_layernorm(w1, b1, gamma, beta)
    return dict(raw=(w1, b1, w2, b2, gamma, beta), folded=pack_geglu(wf, bf), plain=pack_geglu(w1, b1),
                w2=pack_linear(w2), b2=b2.contiguous())


def _x(m, seed=11, pitch=C):
    full = rnd(m, pitch, seed=seed, scale=1.5) + 0.3
    return full.to(BF16)[:, :C]


def _chain(hip, x, wts, ln):
    """The launches the fused operator replaces (on a contiguous copy of the rows)."""
    xc = x.contiguous()
    if ln:
        ones, zeros = torch.ones(C, device=x.device), torch.zeros(C, device=x.device)
        h = hip.layernorm(xc, ones, zeros, 1e-5)
        g = hip.gemm(h, *wts["folded"], act=ACT_GEGLU)
    else:
        g = hip.gemm(xc, *wts["plain"], act=ACT_GEGLU)
    return hip.gemm(g, wts["w2"], wts["b2"], residual=xc)


CASES = [("level-0 rows (81920)", 81920, C), ("one tile", 128, C), ("fewer rows than a tile (50)", 50, C),
         ("partial last tile (1000)", 1000, C), ("three tiles per block and a tail (256 * 300 + 77)", 256 * 300 + 77, C),
         ("rows at pitch 960 (a column block of a wider tensor)", 4096, 960)]


@pytest.mark.parametrize("ln", [True, False], ids=["layernorm", "plain"])
@pytest.mark.parametrize("tag,m,pitch", CASES, ids=[c[0] for c in CASES])
def test_fused_vs_chain_and_emulation(hip, weights, tag, m, pitch, ln):
    x = _x(m, pitch=pitch)
    assert hip.ff_fused_eligible(m, C, HID, ldx=x.stride(0))
    w1, b1 = weights["folded"] if ln else weights["plain"]
    out = hip.ff_geglu_fused(x, w1, b1, weights["w2"], weights["b2"], ln_eps=1e-5 if ln else None)
    torch.cuda.synchronize()
    ref = _chain(hip, x, weights, ln)
    d = (out.float() - ref.float()).abs()
    scale = float(ref.float().abs().max())
    nz = int((d > 0).sum())
    print(f"{tag} / {'LN' if ln else 'plain'}: fused vs three launches: max |d| {float(d.max()):.3e} "
          f"({float(d.max()) / (scale * 2 ** -8):.2f} bf16-ulp of scale), {nz} of {d.numel()} elements differ")
    check(out, ref, f"{tag}: fused vs the three launches", rel=1.5e-3)
    if m <= 4096:
        emu = EmuOps(round_bf16=True, ff_fused_c=C)
        xe = x.cpu()
        want = emu.ff_geglu_fused(xe, w1.cpu(), b1.cpu(), weights["w2"].cpu(), weights["b2"].cpu(), ln_eps=1e-5 if ln else None)
        check(out.cpu(), want, f"{tag}: fused vs emulation")
    again = hip.ff_geglu_fused(x, w1, b1, weights["w2"], weights["b2"], ln_eps=1e-5 if ln else None)
    assert torch.equal(out, again), "repeated launch differs"


def test_fused_vs_fp64_reference_block(hip, weights):
    """x + Linear(GEGLU(LayerNorm(x))) exactly as the reference spells it (attention.py:244-246, 415-442), fp64."""
    w1, b1, w2, b2, gamma, beta = (t.double().cpu() for t in weights["raw"])
    x = _x(3000, seed=21)
    out = hip.ff_geglu_fused(x, *weights["folded"], weights["w2"], weights["b2"], ln_eps=1e-5).double().cpu()
    xd = x.double().cpu()
    h = torch.nn.functional.layer_norm(xd, (C,), gamma, beta, 1e-5) @ w1.t() + b1
    ref = xd + (h[:, :HID] * torch.nn.functional.gelu(h[:, HID:])) @ w2.t() + b2
    err = float((out - ref).norm() / ref.norm())
    print(f"fused feed-forward vs fp64 reference block: rel-L2 {err:.3e}")
    assert err < 6e-3


def test_rows_with_large_mean(hip, weights):
    """Rows with mean 30 and unit variance: the two-pass fp32 statistics in registers against fp64."""
    x = (rnd(2048, C, seed=31, dtype=torch.float32) + 30.0).to(BF16)
    out = hip.ff_geglu_fused(x, *weights["folded"], weights["w2"], weights["b2"], ln_eps=1e-5)
    ref = _chain(hip, x, weights, True)
    check(out, ref, "mean 30 rows: fused vs the three launches", rel=1.5e-3)


def test_eligibility_and_refusals(hip, weights, monkeypatch):
    from tooncrafter_amd._lib import TooncrafterHipError
    assert not hip.ff_fused_eligible(4096, 640, 2560)
    assert not hip.ff_fused_eligible(4096, 320, 2560)
    assert not hip.ff_fused_eligible(4096, 320, 1280, ldx=324)              # rows must start on 16 bytes
    x = _x(256)
    with pytest.raises(ValueError):
        hip.ff_geglu_fused(x, weights["w2"], weights["folded"][1], weights["w2"], weights["b2"])
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.ff_geglu_fused(x.cpu(), *weights["folded"], weights["w2"], weights["b2"])
    monkeypatch.setenv("TC_FF_FUSED", "0")
    assert not hip.ff_fused_eligible(4096, 320, 1280)
    with pytest.raises(TooncrafterHipError):
        hip.ff_geglu_fused(x, *weights["folded"], weights["w2"], weights["b2"], ln_eps=1e-5)


def test_block_routes_through_the_fused_operator(hip, weights, monkeypatch):
    """A level-0 BasicTransformerBlock on the HIP backend, fused feed-forward on vs off."""
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
        real = hip.ff_geglu_fused
        monkeypatch.setattr(hip, "ff_geglu_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            y1 = blk.forward_temporal(x, act)
            monkeypatch.setenv("TC_FF_FUSED", "0")
            y0 = blk.forward_temporal(x, act)
        assert len(calls) == 1
        check(y1, y0, "temporal block, fused feed-forward on vs off", rel=8e-3)   # folded gamma: bf16(w * gamma) vs gamma * bf16 rows
    finally:
        ops.set_backend(prev)
