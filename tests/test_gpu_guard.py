"""Guard-page tests: every operand of every `tc_*` entry point is carved at the very END of its own
hipMalloc'ed region (size a multiple of 2 MiB, the VA right after it kept unmapped where the runtime
allows), at the tiny and ragged shapes of the test models (M of 1..77 rows, H = W = 1, T in {1,3,5},
C = 64, K tails).  A kernel that reads or writes even 16 bytes past an operand then touches an
unmapped page and the GPU raises a memory access fault -- which kills this process -- instead of
silently reading a neighbouring tensor of the caching allocator.  Passing = no out-of-bounds access
at these shapes; values are also checked against the emulated contract.
"""
import ctypes as C

import pytest
import torch

from emu_ops import EmuOps
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"
GRAN = 2 << 20


class _Region:
    """One hipMalloc of k * 2 MiB whose last `nbytes` bytes back a tensor (CUDA array interface)."""
    _hip = None
    adjacent_holes = 0
    regions = 0

    def __init__(self, nbytes, shape, typestr):
        if _Region._hip is None:
            _Region._hip = C.CDLL("libamdhip64.so")
            _Region._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            _Region._hip.hipFree.argtypes = [C.c_void_p]
        hip = _Region._hip
        size = (nbytes + GRAN - 1) // GRAN * GRAN
        base, hole = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(base), size) == 0
        # try to own, then release, the VA right behind the region so that it is certainly unmapped
        if hip.hipMalloc(C.byref(hole), GRAN) == 0:
            if hole.value == base.value + size:
                _Region.adjacent_holes += 1
            hip.hipFree(hole)
        _Region.regions += 1
        self.base, self.size = base.value, size
        self.ptr = base.value + size - nbytes
        assert self.ptr % 16 == 0
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def __del__(self):
        try:
            torch.cuda.synchronize()
            _Region._hip.hipFree(C.c_void_p(self.base))
        except Exception:
            pass


def guard(t: torch.Tensor) -> torch.Tensor:
    """Copy of the (CPU or CUDA) tensor `t` that ends exactly at the end of its own device region."""
    t = t.detach().contiguous()
    nbytes = t.numel() * t.element_size()
    pad = (-nbytes) % 16
    assert pad == 0, "guarded tensors must be a multiple of 16 bytes so they end flush with the region"
    if t.dtype == BF16:
        reg = _Region(nbytes, t.shape, "<i2")
        g = torch.as_tensor(reg, device=DEV).view(BF16)
    else:
        reg = _Region(nbytes, t.shape, {torch.float32: "<f4", torch.uint8: "|u1"}[t.dtype])
        g = torch.as_tensor(reg, device=DEV)
    g._guard_region = reg                       # keep the region alive as long as the tensor
    g.copy_(t.to(DEV))
    assert g.data_ptr() == reg.ptr and g.data_ptr() + nbytes == reg.base + reg.size
    return g


def rnd(*shape, seed=0, scale=1.0, dtype=BF16):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    return guard((torch.randn(*shape, generator=gen) * scale).to(dtype))


def gout(shape, dtype=BF16):
    return guard(torch.zeros(shape, dtype=dtype))


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def close(a, b, what, rel=8e-3):
    torch.cuda.synchronize()
    a, b = a.double().flatten(), b.double().flatten()
    assert torch.isfinite(a).all(), what
    err = float((a - b).norm() / (b.norm() + 1e-30))
    assert err <= rel, f"{what}: rel-L2 {err:.3e}"


def test_guard_regions_end_on_unmapped_memory():
    x = rnd(3, 64)
    torch.cuda.synchronize()
    print(f"guard regions: {_Region.regions}, with a verified free hole right behind: {_Region.adjacent_holes}")
    assert x.data_ptr() % 16 == 0


@pytest.mark.parametrize("m,n,k", [(1, 64, 64), (2, 256, 64), (4, 96, 96), (5, 64, 8), (63, 320, 72), (77, 128, 1024),
                                    (2, 1280, 320), (130, 8, 64), (3, 4, 64)])
def test_guard_gemm_linear(hip, emu, m, n, k):
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5)
    bias = rnd(n, seed=3, dtype=torch.float32) if n % 4 == 0 else None
    vec = n % 8 == 0
    res = rnd(m, n, seed=4) if vec else None
    rb = rnd((m + 1) // 2, n, seed=5, dtype=torch.float32) if vec else None
    f32 = not vec
    out = gout((m, n), torch.float32 if f32 else BF16)
    hip.gemm(a, w, bias, residual=res, row_bias=rb, row_div=2 if rb is not None else 0, act=ACT_SILU, out=out,
             out_f32=f32)
    close(out, emu.gemm(a, w, bias, residual=res, row_bias=rb, row_div=2 if rb is not None else 0, act=ACT_SILU,
                        out_f32=f32), f"guard gemm {m}x{n}x{k}")


def test_guard_gemm_geglu_and_strided_views(hip, emu):
    from tooncrafter_amd.lvdm.common import pack_geglu
    m, c = 5, 64
    x = rnd(m, c, seed=6)
    w = torch.randn(8 * c, c, generator=torch.Generator().manual_seed(7)) * c ** -0.5
    b = torch.randn(8 * c, generator=torch.Generator().manual_seed(8))
    wp, bp = pack_geglu(w, b)
    wp, bp = guard(wp), guard(bp)
    out = gout((m, 4 * c))
    hip.gemm(x, wp, bp, act=ACT_GEGLU, out=out)
    close(out, emu.gemm(x, wp, bp, act=ACT_GEGLU), "guard GEGLU")
    # last-column-block view of a fused buffer (A) and of a wider output: the views end flush with the regions
    big = rnd(7, 3 * 64, seed=9)
    w2 = rnd(64, 64, seed=10, scale=0.125)
    outbuf = gout((7, 128))
    hip.gemm(big[:, 128:], w2, None, out=outbuf[:, 64:])
    close(outbuf[:, 64:], emu.gemm(big[:, 128:], w2, None), "guard strided A/C")


@pytest.mark.parametrize("frames,h,w,cin,cout,stride,ups,pad", [
    (3, 1, 1, 64, 64, 1, False, 1), (5, 1, 1, 128, 256, 1, False, 1), (4, 2, 2, 64, 64, 2, False, 1),
    (3, 1, 1, 64, 64, 1, True, 1), (4, 4, 6, 64, 4, 1, False, 1), (2, 4, 6, 64, 64, 2, False, 0),
    (1, 3, 5, 64, 320, 1, False, 1)])
def test_guard_conv3x3(hip, emu, frames, h, w, cin, cout, stride, ups, pad):
    x = rnd(frames * h * w, cin, seed=11)
    wt = rnd(cout, 9 * cin, seed=12, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=13, dtype=torch.float32)
    if ups:
        ho, wo = 2 * h, 2 * w
    elif pad == 1:
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    else:
        ho, wo = (h + 1 - 3) // stride + 1, (w + 1 - 3) // stride + 1
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, upsample=ups,
                pad=pad)
    f32 = cout % 8 != 0
    out = gout((frames * ho * wo, cout), torch.float32 if f32 else BF16)
    hip.gemm(x, wt, bias, conv=geom, out=out, out_f32=f32)
    close(out, emu.gemm(x, wt, bias, conv=geom, out_f32=f32), f"guard conv3x3 f{frames} {h}x{w} s{stride} ups{ups} pad{pad}")


@pytest.mark.parametrize("b,t,hw,c", [(1, 1, 1, 64), (2, 3, 1, 64), (1, 5, 4, 128), (3, 4, 6, 64)])
def test_guard_convt3(hip, emu, b, t, hw, c):
    x = rnd(b * t * hw, c, seed=14)
    wt = rnd(c, 3 * c, seed=15, scale=(3 * c) ** -0.5)
    bias = rnd(c, seed=16, dtype=torch.float32)
    res = rnd(b * t * hw, c, seed=17)
    geom = dict(kind="t3", frames=b * t, t_len=t, cin=c, h_out=1, w_out=hw)
    out = gout((b * t * hw, c))
    hip.gemm(x, wt, bias, conv=geom, residual=res, out=out)
    close(out, emu.gemm(x, wt, bias, conv=geom, residual=res), f"guard convt3 b{b} t{t} hw{hw}")


def test_guard_gemm_batched(hip, emu):
    f, l, c = 3, 24, 64
    q, k = rnd(f * l, c, seed=18), rnd(f * l, c, seed=19)
    s_h = gout((f * l, l), torch.float32)
    s_e = torch.empty((f * l, l), dtype=torch.float32, device=DEV)
    kw = dict(alpha=c ** -0.5, out_f32=True, batch=f, stride_a=l * c, stride_w=l * c, stride_c=l * l)
    hip.gemm(q[:l], k[:l], out=s_h[:l], **kw)
    emu.gemm(q[:l], k[:l], out=s_e[:l], **kw)
    close(s_h, s_e, "guard batched gemm", rel=1e-4)


@pytest.mark.parametrize("batch,heads,lq,lk,kv_bdiv", [(1, 1, 1, 1, 1), (2, 1, 4, 77, 2), (3, 2, 16, 16, 1),
                                                        (4, 1, 1, 93, 4), (2, 2, 130, 65, 1), (4, 1, 24, 48, 2)])
def test_guard_attention(hip, emu, batch, heads, lq, lk, kv_bdiv):
    c = heads * 64
    kvb = (batch + kv_bdiv - 1) // kv_bdiv
    q, k, v = rnd(batch * lq, c, seed=20), rnd(kvb * lk, c, seed=21), rnd(kvb * lk, c, seed=22)
    out = gout((batch * lq, c))
    hip.attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv, out=out)
    close(out, emu.attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv), "guard attention")
    # K and V as the trailing column blocks of one fused projection (how the model calls it)
    kv = rnd(kvb * lk, 2 * c, seed=23)
    hip.attention(q, kv[:, :c], kv[:, c:], batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv, out=out,
                  accumulate=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("b,t,hw,heads", [(1, 1, 1, 1), (2, 3, 1, 1), (1, 5, 3, 2), (1, 16, 1, 1), (2, 4, 4, 4)])
def test_guard_attention_temporal(hip, emu, b, t, hw, heads):
    qkv = rnd(b * t * hw, 3 * heads * 64, seed=24)
    close(hip.attention_temporal(qkv, b=b, t=t, hw=hw, heads=heads),
          emu.attention_temporal(qkv, b=b, t=t, hw=hw, heads=heads), "guard temporal attention")


@pytest.mark.parametrize("b,hw,c,bias", [(1, 8, 64, False), (1, 8, 128, True), (2, 16, 64, True), (1, 24, 320, False)])
def test_guard_temporal_qkv_attn(hip, emu, b, hw, c, bias):
    """ABI 13 (csrc/qkv_attn.hip): the projection's input rows, the fused weight, the optional bias and the RESULT each end flush
    with their own region -- the gathered row loads (a pixel's 16 frames are hw rows apart), the three weight blocks of a head and
    the per-wave 128-byte row stores touch nothing beyond them.  Also at a wider output pitch (the last column block of a buffer)."""
    heads = c // 64
    x, w = rnd(b * 16 * hw, c, seed=50), rnd(3 * c, c, seed=51, scale=c ** -0.5)
    bq = rnd(3 * c, seed=52, dtype=torch.float32) if bias else None
    kw = dict(b=b, t=16, hw=hw, heads=heads)
    out = gout((b * 16 * hw, c))
    hip.temporal_qkv_attn(x, w, bq, out=out, **kw)
    close(out, emu.temporal_qkv_attn(x, w, bq, **kw), "guard temporal qkv + attention")
    wide = gout((b * 16 * hw, 2 * c))
    hip.temporal_qkv_attn(x, w, bq, out=wide[:, c:], **kw)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, c:], out) and float(wide[:, :c].abs().max()) == 0.0


@pytest.mark.parametrize("samples,rows,c", [(1, 1, 64), (3, 1, 64), (4, 4, 64), (2, 7, 320), (5, 3, 2560), (1, 13, 960)])
def test_guard_groupnorm(hip, emu, samples, rows, c):
    x = rnd(samples * rows, c, seed=25)
    g, b = rnd(c, seed=26, dtype=torch.float32), rnd(c, seed=27, dtype=torch.float32)
    close(hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True),
          emu.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True), "guard groupnorm")


@pytest.mark.parametrize("rows,c", [(1, 64), (5, 64), (3, 320), (17, 640), (2, 1280), (9, 96)])
def test_guard_layernorm(hip, emu, rows, c):
    x = rnd(rows, c, seed=28)
    g, b = rnd(c, seed=29, dtype=torch.float32), rnd(c, seed=30, dtype=torch.float32)
    close(hip.layernorm(x, g, b), emu.layernorm(x, g, b), "guard layernorm")


@pytest.mark.parametrize("samples,rows,c", [(1, 1, 64), (4, 4, 64), (2, 7, 320), (2, 160, 1280), (32, 40, 1280), (5, 3, 2560)])
@pytest.mark.parametrize("pf_bytes", [(16,), (48, 4096), (1 << 20, 80, 65536 + 16, 3 << 20)])
def test_guard_groupnorm_prefetch(hip, emu, samples, rows, c, pf_bytes):
    """ABI 12: the prefetch planes read whole 16-byte units of every listed buffer and not a byte beyond (each buffer ends
    flush with its own region); y is BIT-identical to the plain entry point -- one-launch and three-launch shapes."""
    x = rnd(samples * rows, c, seed=25)
    g, b = rnd(c, seed=26, dtype=torch.float32), rnd(c, seed=27, dtype=torch.float32)
    pf = [rnd(n // 2, seed=40 + i) for i, n in enumerate(pf_bytes)]
    hip.prefetch_min_bytes, keep = 0, hip.prefetch_min_bytes
    try:
        assert len(hip.prefetch_list(x.shape[0], pf)) == len(pf)
        y = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True, prefetch=pf)
    finally:
        hip.prefetch_min_bytes = keep
    y0 = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    close(y, emu.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True), "guard groupnorm + prefetch")


@pytest.mark.parametrize("rows,c", [(1, 64), (3, 320), (17, 640), (2, 1280), (1280, 1280)])
@pytest.mark.parametrize("pf_bytes", [(16,), (1 << 20, 80, 65536 + 16, 3 << 20)])
def test_guard_layernorm_prefetch(hip, emu, rows, c, pf_bytes):
    x = rnd(rows, c, seed=28)
    g, b = rnd(c, seed=29, dtype=torch.float32), rnd(c, seed=30, dtype=torch.float32)
    pf = [rnd(n // 2, seed=50 + i) for i, n in enumerate(pf_bytes)]
    hip.prefetch_min_bytes, keep = 0, hip.prefetch_min_bytes
    try:
        y = hip.layernorm(x, g, b, prefetch=pf)
    finally:
        hip.prefetch_min_bytes = keep
    y0 = hip.layernorm(x, g, b)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    close(y, emu.layernorm(x, g, b), "guard layernorm + prefetch")


def test_prefetch_argument_checks(hip):
    """tc_*_pf: a misaligned or null buffer is an error code, an empty list is the plain call."""
    import ctypes as C
    from tooncrafter_amd import _lib
    x = rnd(8, 64, seed=60)
    g, b = rnd(64, seed=61, dtype=torch.float32), rnd(64, seed=62, dtype=torch.float32)
    y = torch.empty_like(x)
    w = rnd(4096, seed=63)
    stream = torch.cuda.current_stream().cuda_stream
    pf = _lib.TcPrefetch()
    pf.n = 1
    pf.ptr[0], pf.bytes[0] = w.data_ptr() + 2, 64
    assert hip.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), 8, 64, 1e-5, C.byref(pf), stream) == -2
    pf.ptr[0] = None
    assert hip.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), 8, 64, 1e-5, C.byref(pf), stream) == -1
    pf.n = 5
    assert hip.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), 8, 64, 1e-5, C.byref(pf), stream) == -1
    pf.n = 0
    assert hip.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), 8, 64, 1e-5, C.byref(pf), stream) == 0
    assert hip.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), 8, 64, 1e-5, None, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, hip.layernorm(x, g, b))


def test_guard_softmax_layout_elementwise(hip, emu):
    s = rnd(5, 24, seed=31, dtype=torch.float32)
    close(hip.softmax_rows(s, n=21), emu.softmax_rows(s, n=21), "guard softmax")
    x0, x1 = rnd(2, 4, 3, 2, 2, seed=32, dtype=torch.float32), rnd(2, 4, 3, 2, 2, seed=33, dtype=torch.float32)
    rows = hip.nchw_to_rows(x0, x1, c_pad=64)
    assert torch.equal(rows, emu.nchw_to_rows(x0, x1, c_pad=64))
    rg = guard(rows)
    assert torch.equal(hip.rows_to_nchw(rg, c=8, b=2, t=3, h=2, w=2), emu.rows_to_nchw(rows, c=8, b=2, t=3, h=2, w=2))
    rf = rnd(2 * 3 * 4, 4, seed=34, dtype=torch.float32)
    assert torch.equal(hip.rows_to_nchw(rf, c=4, b=2, t=3, h=2, w=2), emu.rows_to_nchw(rf, c=4, b=2, t=3, h=2, w=2))
    rw = rnd(1 * 2 * 3 * 5, 40, seed=38)                     # tiled transpose: 15 pixels per frame, 40 channels
    assert torch.equal(hip.rows_to_nchw(guard(rw), c=40, b=1, t=2, h=3, w=5), emu.rows_to_nchw(rw, c=40, b=1, t=2, h=3, w=5))
    a, b = rnd(3, 64, seed=35), rnd(3, 128, seed=36)
    assert torch.equal(hip.concat_rows(a, b), torch.cat([a, b], 1))
    t = guard(torch.tensor([999.0, 19.0, 0.0, 10.0]))
    close(hip.timestep_embedding(t, 64, 64), emu.timestep_embedding(t, 64, 64), "guard timestep embedding")
    x = rnd(4, 64, seed=37, dtype=torch.float32)
    close(hip.silu_to_bf16(x), emu.silu_to_bf16(x), "guard silu")
    rows3 = rnd(2 * 3 * 4, 4, seed=38, dtype=torch.float32)
    wt, bias = rnd(28, seed=39, dtype=torch.float32)[:27], rnd(4, seed=40, dtype=torch.float32)[:3]
    wt, bias = wt.clone(), bias.clone()          # 27 / 3 floats are not 16-byte multiples: plain tensors
    close(hip.time_mix3(rows3, wt, bias, b=2, t=3, h=2, w_=2), emu.time_mix3(rows3, wt, bias, b=2, t=3, h=2, w_=2),
          "guard time_mix3", rel=1e-5)
    v = rnd(1, 3, 2, 2, 4, seed=41, dtype=torch.float32)
    assert torch.equal(hip.video_to_uint8(v).cpu(), emu.video_to_uint8(v.cpu()))


@pytest.mark.parametrize("three", [False, True])
def test_guard_ddim_step(hip, emu, three):
    shape = (2, 4, 3, 2, 2)
    x, ec, eu, ei, nz = (rnd(*shape, seed=s, dtype=torch.float32) for s in (42, 43, 44, 45, 46))
    sc = dict(sqrt_ac=0.6, sqrt_1m_ac=0.8, sqrt_a_prev=0.7, dir_coef=0.5, sigma=0.3, x0_rescale=0.98)
    kw = dict(e_uncond_img=ei, cfg_img=3.0) if three else {}
    xp, x0 = hip.ddim_step(x, ec, eu, nz, cfg_scale=7.5, guidance_rescale=0.7, **sc, **kw)
    rp, r0 = emu.ddim_step(x, ec, eu, nz, cfg_scale=7.5, guidance_rescale=0.7, **sc, **kw)
    close(xp, rp, "guard ddim x_prev", rel=1e-4)
    close(x0, r0, "guard ddim x0", rel=1e-4)


@pytest.mark.parametrize("m,n,k", [(1, 160, 64), (5, 320, 72), (161, 160, 64)])
def test_guard_gemm_tile16(hip, emu, monkeypatch, m, n, k):
    """The 160x160-tile kernel (gemm16.hip) forced onto tiny / ragged problems whose operands end on unmapped memory."""
    monkeypatch.setenv("TC_GEMM_TILE16", "2")
    a, w = rnd(m, k, seed=50), rnd(n, k, seed=51, scale=k ** -0.5)
    bias, res = rnd(n, seed=52, dtype=torch.float32), rnd(m, n, seed=53)
    out = gout((m, n))
    hip.gemm(a, w, bias, residual=res, out=out)
    close(out, emu.gemm(a, w, bias, residual=res), f"guard tile16 gemm {m}x{n}x{k}")
    x = rnd(3 * 1 * 1, 64, seed=54)
    wt = rnd(160, 9 * 64, seed=55, scale=(9 * 64) ** -0.5)
    geom = dict(kind="3x3", frames=3, cin=64, h_in=1, w_in=1, h_out=1, w_out=1, stride=1, upsample=False)
    out2 = gout((3, 160))
    hip.gemm(x, wt, None, conv=geom, out=out2)
    close(out2, emu.gemm(x, wt, None, conv=geom), "guard tile16 conv 1x1 image")


def test_guard_attention_dual_kv(hip, emu):
    batch, heads, lq, lk, lk2 = 4, 1, 5, 77, 16
    q, k, v = rnd(batch * lq, 64, seed=60), rnd(1 * lk, 64, seed=61), rnd(1 * lk, 64, seed=62)
    k2, v2 = rnd(batch * lk2, 64, seed=63), rnd(batch * lk2, 64, seed=64)
    out = gout((batch * lq, 64))
    kw = dict(batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=4, k2=k2, v2=v2, lk2=lk2, kv2_bdiv=1)
    hip.attention(q, k, v, out=out, **kw)
    close(out, emu.attention(q, k, v, **kw), "guard dual attention")
