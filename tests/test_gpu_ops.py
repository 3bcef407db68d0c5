"""Per-operator parity on the MI355X: every `tc_*` entry point, called through the C ABI
(tooncrafter_amd.ops.HipOps -> ctypes -> libtooncrafter_hip.so), against the plain
PyTorch fp32 statement of the same operator (tests/emu_ops.py) on seeded inputs.

Tolerances: outputs are bf16, so one rounding (2^-9 relative) is the floor; fp32
accumulation order differs between MFMA tiles and torch, so rel-L2 <= 4e-3 and
max-abs <= 3 bf16 ulps of the output scale.  fp32-output operators: rel-L2 <= 1e-4
(inputs are bf16, products exact in fp32, only summation order differs).
"""
import math

import pytest
import torch

from emu_ops import EmuOps
from tooncrafter_amd._lib import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def rnd(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def check(out, ref, what, rel=4e-3, f32=False):
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    o, r = out.double(), ref.double()
    assert torch.isfinite(o).all(), f"{what}: non-finite output"
    err = float((o - r).norm() / (r.norm() + 1e-30))
    mx = float((o - r).abs().max())
    scale = float(r.abs().max()) + 1e-30
    bound = (1e-4 if f32 else rel)
    ulps = mx / (scale * 2 ** -8)
    msg = f"{what}: rel-L2 {err:.3e} max-abs {mx:.3e} (scale {scale:.3e}, {ulps:.2f} bf16-ulp of scale)"
    print(msg)
    assert err <= bound, msg
    if not f32:
        assert ulps <= 4.0, msg


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 320, 320), (1, 1280, 320), (2, 2400, 1280),
                                    (77, 640, 1024), (300, 4, 576), (1000, 96, 96), (4096, 1280, 2560)])
def test_gemm_linear(hip, emu, m, n, k):
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5)
    bias = rnd(n, seed=3, dtype=torch.float32)
    check(hip.gemm(a, w, bias), emu.gemm(a, w, bias), f"gemm {m}x{n}x{k}")


def test_gemm_transpose_detecting(hip, emu):
    """A = I-like with asymmetric B: a swapped C-write (row<->col) cannot pass."""
    n = 256
    a = torch.eye(n, device=DEV, dtype=BF16)
    w = (torch.arange(n, device=DEV)[:, None] * 3 + torch.arange(n, device=DEV)[None, :] % 7).float()
    w = (w / w.max()).to(BF16)
    out = hip.gemm(a, w)
    assert torch.equal(out, w.t().contiguous()), "C-write layout (row/col) is wrong"


def test_gemm_epilogues(hip, emu):
    m, n, k = 640, 320, 640
    a, w = rnd(m, k, seed=4), rnd(n, k, seed=5, scale=k ** -0.5)
    bias = rnd(n, seed=6, dtype=torch.float32)
    res = rnd(m, n, seed=7)
    rb = rnd(5, n + 64, seed=8, dtype=torch.float32)[:, 32:32 + n]          # strided row_bias view
    for act in (ACT_NONE, ACT_SILU, ACT_GELU):
        o = hip.gemm(a, w, bias, act=act, residual=res, row_bias=rb, row_div=128, alpha=0.5, out_scale=0.75)
        r = emu.gemm(a, w, bias, act=act, residual=res, row_bias=rb, row_div=128, alpha=0.5, out_scale=0.75)
        check(o, r, f"gemm epilogue act={act}")
    o = hip.gemm(a, w, bias, out_f32=True)
    check(o, emu.gemm(a, w, bias, out_f32=True), "gemm fp32 out", f32=True)
    # strided A (column slice of a fused buffer) and strided output (write into a wider buffer)
    big = rnd(m, 3 * k, seed=9)
    outbuf = torch.zeros((m, 2 * n), dtype=BF16, device=DEV)
    hip.gemm(big[:, k:2 * k], w, bias, out=outbuf[:, n:])
    check(outbuf[:, n:], emu.gemm(big[:, k:2 * k], w, bias), "gemm strided A / strided C")
    assert float(outbuf[:, :n].abs().max()) == 0.0, "strided output wrote outside its columns"
    # in-place accumulate: residual and out alias (Combiner)
    x = rnd(m, n, seed=10)
    ref = emu.gemm(a, w, bias, residual=x)
    hip.gemm(a, w, bias, residual=x, out=x)
    check(x, ref, "gemm in-place residual")


def test_gemm_geglu(hip, emu):
    from tooncrafter_amd.lvdm.common import pack_geglu
    m, c = 512, 320
    x = rnd(m, c, seed=11)
    w = rnd(8 * c, c, seed=12, scale=c ** -0.5, dtype=torch.float32)
    b = rnd(8 * c, seed=13, dtype=torch.float32)
    wp, bp = pack_geglu(w, b)
    out = hip.gemm(x, wp, bp, act=ACT_GEGLU)
    full = x.float() @ w.to(BF16).float().t() + b
    v, gate = full.chunk(2, dim=-1)
    ref = (v * torch.nn.functional.gelu(gate)).to(BF16)
    check(out, ref, "gemm GEGLU (packed weights) vs unpacked definition")


@pytest.mark.parametrize("frames,h,w,cin,cout,stride,ups", [
    (2, 8, 8, 64, 64, 1, False), (3, 5, 8, 128, 320, 1, False), (2, 10, 16, 64, 128, 2, False),
    (2, 7, 9, 64, 64, 2, False), (2, 5, 8, 128, 128, 1, True), (1, 40, 64, 320, 320, 1, False),
    (2, 6, 6, 64, 3, 1, False)])
def test_gemm_conv3x3(hip, emu, frames, h, w, cin, cout, stride, ups):
    x = rnd(frames * h * w, cin, seed=14)
    wt = rnd(cout, 9 * cin, seed=15, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=16, dtype=torch.float32)
    ho = h * 2 if ups else (h - 1) // stride + 1
    wo = w * 2 if ups else (w - 1) // stride + 1
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, upsample=ups)
    f32 = cout == 3
    o = hip.gemm(x, wt, bias, conv=geom, out_f32=f32)
    r = emu.gemm(x, wt, bias, conv=geom, out_f32=f32)
    check(o, r, f"conv3x3 f{frames} {h}x{w} {cin}->{cout} s{stride} ups{ups}", f32=f32)


def test_gemm_conv3x3_matches_torch_conv2d(hip):
    """The packed-weight convention itself: against F.conv2d on NCHW, not against the emulation."""
    from tooncrafter_amd.lvdm.common import pack_conv3x3
    frames, h, w, cin, cout = 2, 9, 7, 64, 128
    x = rnd(frames, cin, h, w, seed=17)
    wt = rnd(cout, cin, 3, 3, seed=18, scale=(9 * cin) ** -0.5, dtype=torch.float32)
    bias = rnd(cout, seed=19, dtype=torch.float32)
    ref = torch.nn.functional.conv2d(x.float(), wt.to(BF16).float(), bias, padding=1)
    rows = x.permute(0, 2, 3, 1).reshape(frames * h * w, cin).contiguous()
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    out = hip.gemm(rows, pack_conv3x3(wt), bias, conv=geom)
    check(out, ref.permute(0, 2, 3, 1).reshape(frames * h * w, cout).to(BF16), "conv3x3 vs F.conv2d")


@pytest.mark.parametrize("b,t,hw,c", [(1, 16, 40, 64), (2, 4, 64, 128), (1, 3, 24, 320), (2, 1, 16, 64)])
def test_gemm_convt3(hip, emu, b, t, hw, c):
    x = rnd(b * t * hw, c, seed=20)
    wt = rnd(c, 3 * c, seed=21, scale=(3 * c) ** -0.5)
    bias = rnd(c, seed=22, dtype=torch.float32)
    res = rnd(b * t * hw, c, seed=23)
    geom = dict(kind="t3", frames=b * t, t_len=t, cin=c, h_out=1, w_out=hw)
    check(hip.gemm(x, wt, bias, conv=geom, residual=res, out_scale=0.3),
          emu.gemm(x, wt, bias, conv=geom, residual=res, out_scale=0.3), f"convt3 b{b} t{t} hw{hw} c{c}")


@pytest.mark.parametrize("kind", ["linear", "3x3", "3x3s2", "3x3up", "t3"])
def test_gemm_activation_beyond_2gib(hip, kind):
    """An A operand larger than 2 GiB (BASELINE configs[3]: 32 frames of 320x512 through one launch).  Buffer offsets are
    31-bit, so the kernels address A relative to the block's lowest source row (csrc/gemm_common.h: tc_tile_row_lo).
    The rows of the UPPER half (the second 16-frame clip, past the 2 GiB mark) must equal, bit for bit, the same
    launch over that half alone as its own (< 2 GiB) tensor: same tiles, same K order, only the addressing differs."""
    frames, h, w, cin, cout = 32, 320, 512, 256, 64
    if kind == "3x3up":
        h, w = 160, 256
        cin = 1024                                  # source 32 x 160 x 256 x 1024 bf16 = 2.7 GB, output 320 x 512
    rows = frames * h * w
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.empty((rows, cin), dtype=BF16, device=DEV)
    for i in range(0, rows, rows // 8):            # filled in pieces: randn of the whole tensor in fp32 would be 5-10 GB
        x[i:i + rows // 8] = torch.randn((rows // 8, cin), generator=g, device=DEV).to(BF16)
    assert x.numel() * 2 > 2 ** 31 + 2 ** 28
    taps = {"linear": 1, "t3": 3}.get(kind, 9)
    wt = rnd(cout, taps * cin, seed=6, scale=(taps * cin) ** -0.5)
    bias = rnd(cout, seed=7, dtype=torch.float32)

    def geom(fr):
        if kind == "linear":
            return None
        if kind == "t3":
            return dict(kind="t3", frames=fr, t_len=16, cin=cin, h_out=h, w_out=w)
        stride, ups = (2, False) if kind == "3x3s2" else (1, kind == "3x3up")
        ho = h * 2 if ups else (h - 1) // stride + 1
        wo = w * 2 if ups else (w - 1) // stride + 1
        return dict(kind="3x3", frames=fr, cin=cin, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, upsample=ups)
    full = hip.gemm(x, wt, bias, conv=geom(frames))
    half = hip.gemm(x[rows // 2:], wt, bias, conv=geom(frames // 2))
    lo = hip.gemm(x[:rows // 2], wt, bias, conv=geom(frames // 2))
    torch.cuda.synchronize()
    n = full.shape[0] // 2
    assert torch.isfinite(full).all() and float(full[n:].float().abs().mean()) > 0.1
    assert torch.equal(full[:n], lo), f"{kind}: lower half differs"
    assert torch.equal(full[n:], half), f"{kind}: rows past 2 GiB differ"


@pytest.mark.parametrize("m,n,k,kind", [(4096, 320, 320, "linear"), (777, 1280, 64, "linear"), (2048, 640, 1984, "linear"),
                                         (5120, 128, 128, "linear"), (2560, 320, 320, "3x3"), (1280, 1280, 640, "t3"),
                                         (300, 96, 72, "linear")])
def test_gemm_pipelined_loop_equals_plain_loop(hip, m, n, k, kind, monkeypatch):
    """TC_GEMM_PIPE: two K-steps of LDS-DMA in flight (counted vmcnt, raw barriers) against the one-in-flight loop:
    same tiles, same K order -> bit-identical, incl. 1-, 2- and odd-step K loops, K tails, residual / GEGLU epilogues."""
    a_rows = m
    if kind == "3x3":
        geom = dict(kind="3x3", frames=1, cin=k, h_in=40, w_in=64, h_out=40, w_out=64, stride=1, upsample=False)
        kk = 9 * k
    elif kind == "t3":
        geom = dict(kind="t3", frames=8, t_len=4, cin=k, h_out=10, w_out=16)
        kk = 3 * k
    else:
        geom, kk = None, k
    a, w = rnd(a_rows, k, seed=31), rnd(n, kk, seed=32, scale=kk ** -0.5)
    bias, res = rnd(n, seed=33, dtype=torch.float32), rnd(m, n, seed=34)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TC_GEMM_PIPE", mode)
        monkeypatch.setenv("TC_GEMM_TILE16", "0")
        monkeypatch.setenv("TC_GEMM_WS", "0")
        o = [hip.gemm(a, w, bias, conv=geom, residual=res), hip.gemm(a, w, None, conv=geom, act=ACT_SILU, out_scale=0.5)]
        if n % 128 == 0 and geom is None:
            o.append(hip.gemm(a, w, bias, act=ACT_GEGLU))
        torch.cuda.synchronize()
        outs[mode] = o
    for x, y in zip(outs["0"], outs["1"]):
        assert torch.isfinite(x).all() and torch.equal(x, y)


def test_gemm_batched(hip, emu):
    f, l, c = 3, 200, 128
    q, k = rnd(f * l, c, seed=24), rnd(f * l, c, seed=25)
    s_h = torch.empty((f * l, l), dtype=torch.float32, device=DEV)
    s_e = torch.empty_like(s_h)
    kw = dict(alpha=c ** -0.5, out_f32=True, batch=f, stride_a=l * c, stride_w=l * c, stride_c=l * l)
    hip.gemm(q[:l], k[:l], out=s_h[:l], **kw)
    emu.gemm(q[:l], k[:l], out=s_e[:l], **kw)
    check(s_h, s_e, "batched gemm (scores)", f32=True)


# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("batch,heads,lq,lk,kv_bdiv", [
    (2, 2, 128, 128, 1), (3, 5, 160, 160, 1), (2, 1, 40, 40, 1), (4, 2, 64, 77, 2), (4, 2, 100, 16, 1),
    (2, 5, 2560, 2560, 1), (4, 8, 240, 480, 4), (1, 1, 1, 1, 1), (2, 3, 130, 65, 1)])
def test_attention_d64(hip, emu, batch, heads, lq, lk, kv_bdiv):
    c = heads * 64
    kvb = (batch + kv_bdiv - 1) // kv_bdiv
    q = rnd(batch * lq, c, seed=30)
    kv = rnd(kvb * lk, 2 * c, seed=31)
    o = hip.attention(q, kv[:, :c], kv[:, c:], batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv)
    r = emu.attention(q, kv[:, :c], kv[:, c:], batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv)
    check(o, r, f"attention b{batch} h{heads} lq{lq} lk{lk} div{kv_bdiv}", rel=8e-3)


def test_attention_accumulate_and_spike(hip, emu):
    batch, heads, lq, lk = 2, 2, 96, 200
    c = heads * 64
    q, k, v = rnd(batch * lq, c, seed=32), rnd(batch * lk, c, seed=33), rnd(batch * lk, c, seed=34)
    # force the online-softmax rescale: one key far above the rest, late in the sequence
    k[150] = q[7] * 4.0
    base = rnd(batch * lq, c, seed=35)
    o = base.clone()
    hip.attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, out=o, accumulate=True)
    r = base.clone()
    emu.attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, out=r, accumulate=True)
    check(o, r, "attention accumulate + max spike", rel=8e-3)


@pytest.mark.parametrize("b,t,hw,heads", [(1, 16, 40, 5), (2, 4, 64, 1), (1, 16, 7, 8), (2, 3, 5, 2), (1, 1, 9, 1)])
def test_attention_temporal(hip, emu, b, t, hw, heads):
    qkv = rnd(b * t * hw, 3 * heads * 64, seed=36)
    check(hip.attention_temporal(qkv, b=b, t=t, hw=hw, heads=heads),
          emu.attention_temporal(qkv, b=b, t=t, hw=hw, heads=heads), f"temporal attn b{b} t{t} hw{hw} h{heads}",
          rel=8e-3)


# --------------------------------------------------------------------------- norms
@pytest.mark.parametrize("samples,rows,c,silu,eps", [
    (32, 40, 1280, True, 1e-5), (4, 2560, 320, True, 1e-5), (2, 16 * 640, 640, False, 1e-6),
    (3, 100, 64, True, 1e-6), (2, 50, 2560, True, 1e-5), (1, 16 * 2560, 320, True, 1e-5),
    (2, 1000, 128, True, 1e-6), (5, 7, 960, False, 1e-5), (2, 30, 1920, True, 1e-5),
    # level-0 per-frame, level-2 clip-wide, a ragged row count, 640 channels at level-0 rows (round 6: the shapes a 768-thread
    # one-pass instance took before it lost on hardware -- kept as cases of the three-launch path)
    (32, 2560, 320, True, 1e-5), (2, 2560, 1280, True, 1e-5), (16, 2590, 320, False, 1e-6), (8, 2560, 640, True, 1e-5)])
def test_groupnorm(hip, emu, samples, rows, c, silu, eps):
    x = rnd(samples * rows, c, seed=40) * 2.0 + 0.5
    g, b = rnd(c, seed=41, dtype=torch.float32) * 0.1 + 1.0, rnd(c, seed=42, dtype=torch.float32) * 0.1
    check(hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=eps, silu=silu),
          emu.groupnorm(x, g, b, samples=samples, rows=rows, eps=eps, silu=silu),
          f"groupnorm s{samples} r{rows} c{c} silu{silu}")


@pytest.mark.parametrize("rows,c", [(1000, 320), (77, 640), (5, 1280), (333, 512), (64, 64)])
def test_layernorm(hip, emu, rows, c):
    x = rnd(rows, c, seed=43) * 3.0 - 1.0
    g, b = rnd(c, seed=44, dtype=torch.float32) * 0.1 + 1.0, rnd(c, seed=45, dtype=torch.float32) * 0.1
    check(hip.layernorm(x, g, b), emu.layernorm(x, g, b), f"layernorm {rows}x{c}")


def test_softmax_rows(hip, emu):
    s = rnd(300, 2560, seed=46, dtype=torch.float32) * 3.0
    check(hip.softmax_rows(s), emu.softmax_rows(s), "softmax_rows")
    # padded width (K padding of the following GEMM must come out as exact zeros) and the causal text mask
    s2 = rnd(3 * 77, 80, seed=47, dtype=torch.float32) * 2.0
    got = hip.softmax_rows(s2, n=77, causal_period=77)
    check(got, emu.softmax_rows(s2, n=77, causal_period=77), "softmax_rows causal padded")
    assert float(got[:, 77:].abs().max()) == 0.0 and float(got[0, 1:].abs().max()) == 0.0 and float(got[0, 0]) == 1.0
    s3 = rnd(514, 264, seed=48, dtype=torch.float32)
    got = hip.softmax_rows(s3, n=257)
    check(got, emu.softmax_rows(s3, n=257), "softmax_rows padded")
    assert float(got[:, 257:].abs().max()) == 0.0


# --------------------------------------------------------------------------- layout / elementwise
def test_layout_roundtrip(hip, emu):
    x0 = rnd(2, 4, 3, 5, 8, seed=47, dtype=torch.float32)
    x1 = rnd(2, 4, 3, 5, 8, seed=48, dtype=torch.float32)
    rows = hip.nchw_to_rows(x0, x1, c_pad=64, scale=0.5)
    assert torch.equal(rows, emu.nchw_to_rows(x0, x1, c_pad=64, scale=0.5))
    back = hip.rows_to_nchw(rows, c=8, b=2, t=3, h=5, w=8)
    assert torch.equal(back, emu.rows_to_nchw(rows, c=8, b=2, t=3, h=5, w=8))
    rf = rnd(2 * 3 * 40, 4, seed=49, dtype=torch.float32)
    assert torch.equal(hip.rows_to_nchw(rf, c=4, b=2, t=3, h=5, w=8), emu.rows_to_nchw(rf, c=4, b=2, t=3, h=5, w=8))
    a, b = rnd(1000, 320, seed=50), rnd(1000, 640, seed=51)
    assert torch.equal(hip.concat_rows(a, b), torch.cat([a, b], 1))


@pytest.mark.parametrize("b,t,h,w,c,ld", [(1, 2, 5, 9, 128, 128), (2, 1, 8, 8, 96, 96), (1, 3, 7, 11, 40, 64), (1, 1, 16, 20, 512, 512)])
def test_rows_to_nchw_wide_rows_tiled(hip, emu, b, t, h, w, c, ld):
    """bf16 rows of >= 32 channels take the LDS-tiled transpose: ragged pixel / channel tiles, column-sliced source."""
    wide = rnd(b * t * h * w, ld, seed=40)
    rows = wide[:, :c]
    assert torch.equal(hip.rows_to_nchw(rows, c=c, b=b, t=t, h=h, w=w), emu.rows_to_nchw(rows, c=c, b=b, t=t, h=h, w=w))


def test_embedding_helpers(hip, emu):
    t = torch.tensor([999.0, 19.0, 0.0, 10.0], device=DEV)
    check(hip.timestep_embedding(t, 320, 320), emu.timestep_embedding(t, 320, 320), "timestep_embedding")
    x = rnd(4, 1280, seed=52, dtype=torch.float32)
    check(hip.silu_to_bf16(x), emu.silu_to_bf16(x), "silu_to_bf16")
    ti = torch.tensor([999, 19, 0, 10, 601], device=DEV, dtype=torch.int64)        # ABI 9: the samplers' int64 timesteps
    assert torch.equal(hip.timestep_embedding(ti, 320, 320), hip.timestep_embedding(ti.to(torch.float32), 320, 320))


@pytest.mark.parametrize("rows,c,n,dtype", [(1, 8, 2, BF16), (2560, 320, 2, BF16), (777, 328, 3, BF16), (2, 4700, 3, torch.float32)])
def test_repeat_rows(hip, rows, c, n, dtype):
    """ABI 9: x.repeat(n, 1) on 16-byte rows (the shared prefix of batched guidance leaving its single copy)."""
    x = rnd(rows, c, seed=57, dtype=dtype)
    out = hip.repeat_rows(x, n)
    assert out.shape == (n * rows, c) and torch.equal(out, x.repeat(n, 1))


def test_time_mix3(hip, emu):
    b, t, h, w = 2, 5, 6, 8
    rows = rnd(b * t * h * w, 3, seed=53, dtype=torch.float32)
    wt, bias = rnd(27, seed=54, dtype=torch.float32), rnd(3, seed=55, dtype=torch.float32)
    check(hip.time_mix3(rows, wt, bias, b=b, t=t, h=h, w_=w), emu.time_mix3(rows, wt, bias, b=b, t=t, h=h, w_=w),
          "time_mix3", f32=True)


@pytest.mark.parametrize("cfg,resc,noise", [(7.5, 0.7, True), (7.5, 0.0, False), (1.0, 0.0, True)])
def test_ddim_step(hip, emu, cfg, resc, noise):
    shape = (2, 4, 16, 40, 64)
    x, ec, eu = (rnd(*shape, seed=s, dtype=torch.float32) for s in (56, 57, 58))
    nz = rnd(*shape, seed=59, dtype=torch.float32) if noise else None
    sc = dict(sqrt_ac=0.6, sqrt_1m_ac=0.8, sqrt_a_prev=0.7, dir_coef=0.5, sigma=0.3 if noise else 0.0, x0_rescale=0.98)
    eu_arg = eu if cfg != 1.0 else None
    xp, x0 = hip.ddim_step(x, ec, eu_arg, nz, cfg_scale=cfg, guidance_rescale=resc, **sc)
    rp, r0 = emu.ddim_step(x, ec, eu_arg, nz, cfg_scale=cfg, guidance_rescale=resc, **sc)
    check(xp, rp, f"ddim x_prev cfg{cfg} resc{resc}", f32=True)
    check(x0, r0, f"ddim pred_x0 cfg{cfg} resc{resc}", f32=True)


def test_gemm_split_k_low_resolution_layers(hip, emu):
    """The lowest-resolution UNet layers (M = B*T*5*8 = 1280 rows) run split-K: partial fp32 tiles in the
    workspace, fixed-order reduction + epilogue in a second kernel.  Parity with every fused term, and
    bit-identical reruns."""
    from tooncrafter_amd._lib import TcGemmParams
    import ctypes as C
    m, n, cin = 1280, 1280, 1280
    x = rnd(m, cin, seed=70)
    w = rnd(n, 9 * cin, seed=71, scale=(9 * cin) ** -0.5)
    bias, rb = rnd(n, seed=72, dtype=torch.float32), rnd(32, n, seed=73, dtype=torch.float32)
    res = rnd(m, n, seed=74)
    geom = dict(kind="3x3", frames=32, cin=cin, h_in=5, w_in=8, h_out=5, w_out=8, stride=1, upsample=False)
    p = TcGemmParams(); p.m, p.n, p.k, p.batch, p.act = m, n, 9 * cin, 1, 0
    assert hip.lib.tc_gemm_workspace(C.byref(p)) == 4 * m * n * 4            # 100 tiles -> 4 K-slices (a power of two)
    got = hip.gemm(x, w, bias, conv=geom, row_bias=rb, row_div=40, residual=res, act=ACT_SILU)
    ref = emu.gemm(x, w, bias, conv=geom, row_bias=rb, row_div=40, residual=res, act=ACT_SILU)
    check(got, ref, "split-K conv3x3 L3 (bias + row_bias + silu + residual)")
    assert torch.equal(got, hip.gemm(x, w, bias, conv=geom, row_bias=rb, row_div=40, residual=res, act=ACT_SILU))
    # linear with a ragged K tail and fp32 output (K just past the split threshold)
    a2, w2 = rnd(1000, 8200, seed=75), rnd(264, 8200, seed=76, scale=8200 ** -0.5)
    p.m, p.n, p.k = 1000, 264, 8200
    assert hip.lib.tc_gemm_workspace(C.byref(p)) > 0
    check(hip.gemm(a2, w2, out_f32=True), emu.gemm(a2, w2, out_f32=True), "split-K linear ragged f32", f32=True)
    p.k = 5120
    assert hip.lib.tc_gemm_workspace(C.byref(p)) == 0                         # short K: 64x64 tiles instead


@pytest.mark.parametrize("m,n,k", [(1280, 1280, 1280), (1000, 264, 704), (130, 3840, 1280), (2048, 200, 320), (77, 64, 64)])
def test_gemm_w_stationary_walk_is_the_same_gemm(hip, emu, monkeypatch, m, n, k):
    """TC_GEMM_NMAJOR=2 forces the W-stationary tile walk (N-major tiles dealt to the XCDs in 8 contiguous runs,
    surplus blocks exit): ragged tile counts, tile counts below 8, and the default walk must agree bit for bit."""
    a, w = rnd(m, k, seed=90), rnd(n, k, seed=91, scale=k ** -0.5)
    bias, res = rnd(n, seed=92, dtype=torch.float32), rnd(m, n, seed=93)
    monkeypatch.setenv("TC_GEMM_NMAJOR", "0")
    base = hip.gemm(a, w, bias, residual=res)
    monkeypatch.setenv("TC_GEMM_NMAJOR", "2")
    walk = hip.gemm(a, w, bias, residual=res)
    assert torch.equal(base, walk)
    check(walk, emu.gemm(a, w, bias, residual=res), f"W-stationary walk {m}x{n}x{k}")


def test_gemm_chunked_tile_order_for_wide_n(hip, emu):
    """Layers whose weight matrix outgrows one XCD's L2 (N*K*2 > 4 MiB, >= 16 N-tiles) walk their tiles in
    chunks of 8 N-tiles x all M-tiles of the XCD: a different, still bijective block -> tile map.  Ragged M,
    an N-tile count that is not a multiple of 8, both tile families."""
    a, w, b = rnd(1100, 1024, seed=80), rnd(2680, 1024, seed=81, scale=1024 ** -0.5), rnd(2680, seed=82, dtype=torch.float32)
    check(hip.gemm(a, w, b, act=ACT_SILU), emu.gemm(a, w, b, act=ACT_SILU), "chunked order 128x128, 21 N-tiles")
    a, w, b = rnd(700, 1280, seed=83), rnd(5120, 1280, seed=84, scale=1280 ** -0.5), rnd(5120, seed=85, dtype=torch.float32)
    check(hip.gemm(a, w, b, act=ACT_GEGLU), emu.gemm(a, w, b, act=ACT_GEGLU), "chunked order 256x320 GEGLU, 16 N-tiles")
    a, w = rnd(2100, 640, seed=86), rnd(5120, 640, seed=87, scale=640 ** -0.5)
    check(hip.gemm(a, w, None, act=ACT_GEGLU), emu.gemm(a, w, None, act=ACT_GEGLU), "chunked order 128x128 GEGLU, 40 N-tiles")


@pytest.mark.parametrize("resc,cfg_img", [(0.7, 3.0), (0.0, None)])
def test_ddim_step_three_way_guidance(hip, emu, resc, cfg_img):
    """Row f3: e_uncond + cfg_img (e_uncond_img - e_uncond) + s (e_cond - e_uncond_img), then rescale."""
    shape = (1, 4, 16, 40, 64)
    x, ec, eu, ei, nz = (rnd(*shape, seed=s, dtype=torch.float32) for s in (60, 61, 62, 63, 64))
    sc = dict(sqrt_ac=0.6, sqrt_1m_ac=0.8, sqrt_a_prev=0.7, dir_coef=0.5, sigma=0.3, x0_rescale=0.98)
    xp, x0 = hip.ddim_step(x, ec, eu, nz, cfg_scale=7.5, guidance_rescale=resc, e_uncond_img=ei, cfg_img=cfg_img, **sc)
    rp, r0 = emu.ddim_step(x, ec, eu, nz, cfg_scale=7.5, guidance_rescale=resc, e_uncond_img=ei, cfg_img=cfg_img, **sc)
    check(xp, rp, f"ddim3 x_prev resc{resc} cfg_img{cfg_img}", f32=True)
    check(x0, r0, f"ddim3 pred_x0 resc{resc} cfg_img{cfg_img}", f32=True)
    # the third pass is really used; with cfg_img == cfg_scale (the reference's default) the formula
    # collapses algebraically to two-way guidance
    xp2, _ = hip.ddim_step(x, ec, eu, nz, cfg_scale=7.5, guidance_rescale=resc, **sc)
    if cfg_img is not None:
        assert (xp2 - xp).abs().max() > 1e-2
    else:
        assert (xp2 - xp).abs().max() < 1e-4


def test_video_to_uint8_is_bit_exact(hip, emu):
    """Row f4: bytes identical to clamp -> (x+1)/2 -> *255 -> uint8 -> permute of inference.py:148-153."""
    g = torch.Generator().manual_seed(65)
    v = (torch.randn(2, 3, 5, 24, 40, generator=g) * 0.8)
    v[0, :, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, -0.999999, 0.999999, 0.00392156])
    got = hip.video_to_uint8(v.to(DEV))
    ref = emu.video_to_uint8(v)                                   # the reference arithmetic on the CPU
    assert got.dtype == torch.uint8 and tuple(got.shape) == (2, 5, 24, 40, 3)
    assert torch.equal(got.cpu(), ref)
    # every multiple of 1/255 round-trips (boundary values of the truncation)
    lv = (torch.arange(256, dtype=torch.float32) / 255.0 * 2.0 - 1.0).reshape(1, 1, 1, 16, 16).repeat(1, 3, 1, 1, 1)
    assert torch.equal(hip.video_to_uint8(lv.to(DEV)).cpu(), emu.video_to_uint8(lv))


# --------------------------------------------------------------------------- error behaviour
def test_bad_arguments_raise(hip):
    from tooncrafter_amd._lib import TooncrafterHipError
    a, w = rnd(64, 64), rnd(64, 64)
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.gemm(a.cpu(), w)                               # CPU tensor: product path is GPU-only
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.gemm(a[:, :60], w[:, :60].contiguous())        # K not a multiple of 8
    with pytest.raises((TooncrafterHipError, ValueError)):
        hip.groupnorm(rnd(10, 48), torch.ones(48, device=DEV), torch.zeros(48, device=DEV), samples=1, rows=10, eps=1e-5)


# --------------------------------------------------------------------------- wide-tile GEMM (gemm_wide.hip)
@pytest.mark.parametrize("m,n,k,kw", [
    (65536, 320, 320, dict(res=True)),                 # BN=320, one N tile, 256 blocks
    (16384, 960, 320, dict()),                         # BN=320, three N tiles
    (16500, 960, 328, dict(res=True, rb=True)),        # ragged M and K tails
    (51200, 256, 512, dict(res=True, f32=True)),       # BN=256
    (50000, 128, 192, dict(act=ACT_SILU)),             # BN=128
    (24576, 640, 2560, dict(res=True)),                # long K
    (12288, 1536, 512, dict()),                        # N not a multiple of 320 -> BN=256, N tail (1536 = 6*256)
    (49152, 416, 64, dict()),                          # N tail inside the last BN=256 tile
])
def test_gemm_wide_linear(hip, emu, m, n, k, kw):
    a, w = rnd(m, k, seed=60), rnd(n, k, seed=61, scale=k ** -0.5)
    bias = rnd(n, seed=62, dtype=torch.float32)
    res = rnd(m, n, seed=63) if kw.get("res") else None
    rb = rnd((m + 4095) // 4096, n, seed=64, dtype=torch.float32) if kw.get("rb") else None
    args = dict(act=kw.get("act", ACT_NONE), residual=res, row_bias=rb, row_div=4096 if rb is not None else 0,
                out_f32=kw.get("f32", False), alpha=0.9, out_scale=1.1)
    check(hip.gemm(a, w, bias, **args), emu.gemm(a, w, bias, **args), f"wide gemm {m}x{n}x{k} {kw}",
          f32=kw.get("f32", False))


def test_gemm_wide_geglu(hip):
    from tooncrafter_amd.lvdm.common import pack_geglu
    for m, c in ((16384, 320), (6144, 640)):
        x = rnd(m, c, seed=65)
        w = rnd(8 * c, c, seed=66, scale=c ** -0.5, dtype=torch.float32)
        b = rnd(8 * c, seed=67, dtype=torch.float32)
        wp, bp = pack_geglu(w, b)
        out = hip.gemm(x, wp, bp, act=ACT_GEGLU)
        full = x.float() @ w.to(BF16).float().t() + b
        v, gate = full.chunk(2, dim=-1)
        check(out, (v * torch.nn.functional.gelu(gate)).to(BF16), f"wide GEGLU m{m} c{c}")


@pytest.mark.parametrize("frames,h,w,cin,cout,t3", [
    (32, 40, 64, 320, 320, False), (32, 40, 64, 64, 320, False), (16, 80, 128, 128, 256, False),
    (32, 40, 64, 320, 320, True), (16, 64, 64, 128, 128, True)])
def test_gemm_wide_conv(hip, emu, frames, h, w, cin, cout, t3):
    x = rnd(frames * h * w, cin, seed=68)
    taps = 3 if t3 else 9
    wt = rnd(cout, taps * cin, seed=69, scale=(taps * cin) ** -0.5)
    bias = rnd(cout, seed=70, dtype=torch.float32)
    res = rnd(frames * h * w, cout, seed=71)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    check(hip.gemm(x, wt, bias, conv=geom, residual=res), emu.gemm(x, wt, bias, conv=geom, residual=res),
          f"wide conv t3={t3} f{frames} {h}x{w} {cin}->{cout}")


# --------------------------------------------------------------------------- 160x160-tile GEMM (gemm16.hip)
@pytest.mark.parametrize("m,n,k,kw", [
    (20480, 640, 640, dict(res=True)),                 # level-1 projection: 512 tiles = one round (heuristic picks it)
    (81920, 320, 320, dict(res=True, rb=True)),        # level-0 projection: 1024 tiles
    (1, 160, 64, dict()),                              # one row
    (77, 320, 328, dict(res=True, act=ACT_SILU)),      # ragged M and K tails
    (1000, 480, 96, dict(f32=True)),                   # fp32 output, three column tiles
    (4153, 640, 1280, dict(rb=True, act=ACT_GELU)),    # ragged M over many tiles
    (163, 160, 2048, dict(res=True)),                  # M just past one tile
])
def test_gemm_tile16_linear(hip, emu, monkeypatch, m, n, k, kw):
    monkeypatch.setenv("TC_GEMM_TILE16", "2")
    a, w = rnd(m, k, seed=90), rnd(n, k, seed=91, scale=k ** -0.5)
    bias = rnd(n, seed=92, dtype=torch.float32)
    res = rnd(m, n, seed=93) if kw.get("res") else None
    rb = rnd((m + 511) // 512, n, seed=94, dtype=torch.float32) if kw.get("rb") else None
    args = dict(act=kw.get("act", ACT_NONE), residual=res, row_bias=rb, row_div=512 if rb is not None else 0,
                out_f32=kw.get("f32", False), alpha=0.9, out_scale=1.1)
    got = hip.gemm(a, w, bias, **args)
    monkeypatch.setenv("TC_GEMM_TILE16", "0")
    other = hip.gemm(a, w, bias, **args)                  # the 128x128 / 256-row kernels on the same problem
    check(got, emu.gemm(a, w, bias, **args), f"tile16 gemm {m}x{n}x{k} {kw}", f32=kw.get("f32", False))
    check(got, other, f"tile16 vs 128-tile kernels {m}x{n}x{k}", f32=kw.get("f32", False))


def test_gemm_tile16_transpose_detecting(hip, monkeypatch):
    monkeypatch.setenv("TC_GEMM_TILE16", "2")
    n = 320
    a = torch.eye(n, device=DEV, dtype=BF16)
    w = (torch.arange(n, device=DEV)[:, None] * 3 + torch.arange(n, device=DEV)[None, :] % 7).float()
    w = (w / w.max()).to(BF16)
    assert torch.equal(hip.gemm(a, w), w.t().contiguous()), "tile16 C-write layout (row/col) is wrong"


@pytest.mark.parametrize("frames,h,w,cin,cout,stride,ups,t3", [
    (32, 40, 64, 320, 320, 1, False, False), (3, 5, 8, 128, 320, 1, False, False), (2, 10, 16, 64, 160, 2, False, False),
    (2, 5, 8, 128, 160, 1, True, False), (32, 20, 32, 640, 640, 1, False, True), (6, 3, 5, 64, 160, 1, False, True)])
def test_gemm_tile16_conv(hip, emu, monkeypatch, frames, h, w, cin, cout, stride, ups, t3):
    monkeypatch.setenv("TC_GEMM_TILE16", "2")
    x = rnd(frames * h * w, cin, seed=95)
    taps = 3 if t3 else 9
    wt = rnd(cout, taps * cin, seed=96, scale=(taps * cin) ** -0.5)
    bias = rnd(cout, seed=97, dtype=torch.float32)
    ho = h * 2 if ups else (h - 1) // stride + 1
    wo = w * 2 if ups else (w - 1) // stride + 1
    geom = dict(kind="t3", frames=frames, t_len=frames if frames < 16 else 16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, upsample=ups)
    res = rnd(frames * ho * wo, cout, seed=98) if not t3 else rnd(frames * h * w, cout, seed=98)
    check(hip.gemm(x, wt, bias, conv=geom, residual=res), emu.gemm(x, wt, bias, conv=geom, residual=res),
          f"tile16 conv t3={t3} f{frames} {h}x{w} {cin}->{cout} s{stride} ups{ups}")


# --------------------------------------------------------------------------- two key/value sets, two softmaxes, one launch
@pytest.mark.parametrize("batch,heads,lq,lk,div,lk2,div2", [
    (32, 5, 2560, 77, 16, 16, 1),      # level-0 cross-attention: 77 text keys per clip, 16 image keys per frame
    (8, 2, 100, 77, 4, 16, 1), (4, 1, 33, 77, 4, 64, 4), (2, 3, 130, 200, 1, 65, 2), (3, 1, 1, 1, 1, 1, 1)])
def test_attention_dual_kv(hip, emu, batch, heads, lq, lk, div, lk2, div2):
    c = heads * 64
    q = rnd(batch * lq, c, seed=110)
    kv = rnd(((batch + div - 1) // div) * lk, 2 * c, seed=111)
    kv2 = rnd(((batch + div2 - 1) // div2) * lk2, 2 * c, seed=112)
    kw = dict(batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=div)
    kw2 = dict(k2=kv2[:, :c], v2=kv2[:, c:], lk2=lk2, kv2_bdiv=div2)
    got = hip.attention(q, kv[:, :c], kv[:, c:], **kw, **kw2)
    check(got, emu.attention(q, kv[:, :c], kv[:, c:], **kw, **kw2), f"dual attention b{batch} h{heads} lq{lq} lk{lk}+{lk2}",
          rel=8e-3)
    # against the two-launch form it replaces (second launch accumulates into the bf16 result of the first)
    two = hip.attention(q, kv[:, :c], kv[:, c:], **kw)
    hip.attention(q, kv2[:, :c], kv2[:, c:], batch=batch, heads=heads, lq=lq, lk=lk2, kv_bdiv=div2, out=two, accumulate=True)
    check(got, two, "dual attention vs two launches", rel=8e-3)


def test_groupnorm_large_mean_two_pass(hip):
    """Activations with a large common offset (mean 60, spread 1): the variance must come out of a true two-pass
    sum((x - mean)^2) or an equally careful accumulation, not of E[x^2] - mean^2 in low precision.  Against float64."""
    for samples, rows, c in ((4, 160, 1280), (2, 2560, 320), (2, 10240, 640), (32, 2560, 320)):   # single-pass (two sizes), 3-launch path (two sizes)
        gen = torch.Generator().manual_seed(77)
        x = (torch.randn(samples * rows, c, generator=gen) + 60.0).to(BF16).to(DEV)
        g = torch.ones(c, device=DEV)
        b = torch.zeros(c, device=DEV)
        got = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=False)
        xd = x.double().reshape(samples, rows, 32, c // 32).permute(0, 2, 1, 3)
        mean = xd.mean(dim=(2, 3), keepdim=True)
        var = ((xd - mean) ** 2).mean(dim=(2, 3), keepdim=True)
        ref = ((xd - mean) / (var + 1e-5).sqrt()).permute(0, 2, 1, 3).reshape(samples * rows, c)
        err = float((got.double() - ref).norm() / ref.norm())
        print(f"groupnorm mean 60 +- 1, s{samples} r{rows} c{c}: rel-L2 vs float64 {err:.3e}")
        assert err < 6e-3, err          # bf16 output rounding is 2e-3..4e-3 of a unit-variance result
