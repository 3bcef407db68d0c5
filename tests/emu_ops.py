"""CPU emulation of the operator contract of tooncrafter_amd.ops.HipOps.

TEST DOUBLE ONLY (lives under tests/, never imported by the product).  It states,
in plain PyTorch, what each `tc_*` entry point must compute -- bf16 storage, fp32
arithmetic -- so that (a) the host logic of the module mirror (layouts, weight
packing, state-dict mapping, sampler scalars) can be validated against the oracle
and the reference goldens without a GPU, and (b) the GPU tests have a per-operator
specification to check the HIP kernels against.
"""
import torch
import torch.nn.functional as F

from tooncrafter_amd._lib import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU

BF16 = torch.bfloat16


def _f(t):
    return t.to(torch.float32)


class EmuOps:
    name = "emu"

    def __init__(self, round_bf16=True, ln_fusion_k=None, gn_rows=0, ff_fused_c=None, tb_fused_c=None, tqa=False):
        self.round = round_bf16
        self.ff_fused_c = ff_fused_c         # tests only: width whose feed-forward takes the one-launch route (the library: 320)
        self.ff_fused_calls = 0
        self.tb_fused_c = tb_fused_c         # tests only: width whose temporal self-attention takes the one-launch route (library: 320)
        self.tb_fused_calls = 0
        self.tqa = tqa                       # tests only: offer the qkv + attention launch (ABI 13; library: wherever c = heads * 64, t = 16, hw % 8 == 0)
        self.tqa_calls = 0
        self.ln_fusion_k = ln_fusion_k       # tests only: accept a_norm_eps for EVERY consumer with this K.  The HIP library's
                                             # default rule (csrc/gemm_ws.hip: ws_shape_ok, mode 1) is narrower: K = 320, N = 320,
                                             # no GEGLU, M >= 65536 -- only the level-0 projections; TC_GEMM_WS=2 widens it to the
                                             # qkv / GEGLU consumers (tests/test_gpu_gemm_ws.py runs those)
        self.ln_fused_calls = 0
        self.gn_rows = gn_rows               # tests only: row-block height of the emulated producer statistics (0 = none)
        self.gn_part_made = self.gn_part_used = 0

    def gemm_ln_eligible(self, m, n, k, *, geglu=False, lda=None):
        return self.ln_fusion_k is not None and k == self.ln_fusion_k

    def ff_fused_eligible(self, m, c, hidden, *, ldx=None):
        return self.ff_fused_c is not None and c == self.ff_fused_c and hidden == 4 * c

    def ff_geglu_fused(self, x, w1, b1, w2, b2, *, ln_eps=None):
        """csrc/ff_fused.hip: the same roundings as the three launches it replaces (normalised rows and hidden values in
        bf16, fp32 sums), so the mirror IS the two emulated GEMMs."""
        self.ff_fused_calls += 1
        g = self.gemm(x, w1, b1, act=ACT_GEGLU, a_norm_eps=ln_eps) if ln_eps is not None else self.gemm(x, w1, b1, act=ACT_GEGLU)
        return self.gemm(g, w2, b2, residual=x)

    def temporal_attn_fused_eligible(self, *, b, t, hw, c, heads, ldx=None):
        return self.tb_fused_c is not None and c == self.tb_fused_c and heads * 64 == c

    def temporal_attn_fused(self, x, wqkv, bqkv, wo, bo, *, b, t, hw, heads, ln_eps=None, scale=None):
        """csrc/tb_fused.hip: the roundings of the launches it replaces (normalised rows, q / k / v and the attention output
        in bf16, fp32 sums; its softmax weights are bf16 where tc_attn_temporal keeps fp32 -- inside the operator bound)."""
        self.tb_fused_calls += 1
        qkv = self.gemm(x, wqkv, bqkv, a_norm_eps=ln_eps) if ln_eps is not None else self.gemm(x, wqkv, bqkv)
        a = self.attention_temporal(qkv, b=b, t=t, hw=hw, heads=heads, scale=scale)
        return self.gemm(a, wo, bo, residual=x)

    def temporal_qkv_attn_eligible(self, *, b, t, hw, c, heads, ldx=None):
        return bool(self.tqa) and t == 16 and heads * 64 == c and hw % 8 == 0

    def temporal_qkv_attn(self, x, wqkv, bqkv=None, *, b, t, hw, heads, scale=None, out=None):
        """csrc/qkv_attn.hip: the roundings of the two launches it replaces (q / k / v and the attention output in bf16, fp32
        sums; its softmax weights are bf16 where tc_attn_temporal keeps fp32 -- inside the operator bound)."""
        self.tqa_calls += 1
        y = self.attention_temporal(self.gemm(x, wqkv, bqkv), b=b, t=t, hw=hw, heads=heads, scale=scale)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def _out(self, x, f32=False):
        if f32:
            return x.to(torch.float32)
        return x.to(BF16) if self.round else x

    # ------------------------------------------------------------------ GEMM family
    def _gather(self, a, conv, k):
        if conv is None:
            return _f(a[:, :k])
        fr, cin = conv["frames"], conv["cin"]
        ho, wo = conv["h_out"], conv["w_out"]
        if conv["kind"] == "3x3":
            hi, wi = conv.get("h_in", ho), conv.get("w_in", wo)
            x = _f(a[:fr * hi * wi, :cin]).reshape(fr, hi, wi, cin).permute(0, 3, 1, 2)
            if conv.get("upsample", False):
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            if conv.get("pad", 1) == 0:
                cols = F.unfold(F.pad(x, (0, 1, 0, 1)), kernel_size=3, padding=0, stride=conv.get("stride", 1))
            else:
                cols = F.unfold(x, kernel_size=3, padding=1, stride=conv.get("stride", 1))   # [fr, cin*9, L]
            cols = cols.reshape(fr, cin, 9, ho * wo).permute(0, 3, 2, 1)                 # [fr, L, tap, cin]
            return cols.reshape(fr * ho * wo, 9 * cin)
        t = conv["t_len"]
        hw = ho * wo
        x = _f(a[:fr * hw, :cin]).reshape(fr // t, t, hw, cin)
        xp = F.pad(x, (0, 0, 0, 0, 1, 1))
        cols = torch.stack([xp[:, 0:t], xp[:, 1:t + 1], xp[:, 2:t + 2]], dim=3)           # [b, t, hw, tap, cin]
        return cols.reshape(fr * hw, 3 * cin)

    def gemm(self, a, w, bias=None, *, act=ACT_NONE, residual=None, row_bias=None, row_div=0, alpha=1.0,
             out_scale=1.0, out=None, out_f32=False, conv=None, batch=1, stride_a=0, stride_w=0, stride_c=0,
             m=None, a_norm_eps=None, gn_stats=False):
        if gn_stats:
            # ABI 9: the producer's partial GroupNorm sums, per block of `gn_rows` rows (tests: 160, like the 160-tile kernel)
            res = self.gemm(a, w, bias, act=act, residual=residual, row_bias=row_bias, row_div=row_div, alpha=alpha,
                            out_scale=out_scale, out=out, out_f32=out_f32, conv=conv, batch=batch, stride_a=stride_a,
                            stride_w=stride_w, stride_c=stride_c, m=m, a_norm_eps=a_norm_eps)
            if self.gn_rows <= 0 or act == ACT_GEGLU or out_f32 or not res.is_contiguous():
                return res, None
            from tooncrafter_amd.ops import GnPart
            r = self.gn_rows
            x = _f(res)
            nb = (x.shape[0] + r - 1) // r
            xp = F.pad(x, (0, 0, 0, nb * r - x.shape[0])).reshape(nb, r, x.shape[1])
            self.gn_part_made += 1
            return res, GnPart(torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous(), r, res)
        n, k = w.shape
        if batch > 1:
            assert conv is None and residual is None and row_bias is None
            mm = a.shape[0]
            res = []
            for i in range(batch):
                ai = torch.as_strided(a, a.shape, a.stride(), a.storage_offset() + i * stride_a)
                wi = torch.as_strided(w, w.shape, w.stride(), w.storage_offset() + i * stride_w)
                oi = torch.as_strided(out, out.shape, out.stride(), out.storage_offset() + i * stride_c)
                self.gemm(ai, wi, bias, act=act, alpha=alpha, out_scale=out_scale, out=oi, out_f32=out_f32)
            return out
        A = self._gather(a, conv, k)
        if m is not None:
            A = A[:m]
        if a_norm_eps is not None:
            # ABI 8: rows normalised over K (fp32 statistics, biased variance) and rounded to bf16 -- the tile the MFMAs
            # read; the affine half of the LayerNorm is already inside w / bias (common.fold_layernorm)
            assert conv is None
            mu = A.mean(1, keepdim=True)
            A = (A - mu) * torch.rsqrt(((A - mu) ** 2).mean(1, keepdim=True) + a_norm_eps)
            A = _f(A.to(BF16)) if self.round else A
            self.ln_fused_calls += 1
        acc = (A @ _f(w).t()) * alpha
        if act == ACT_GEGLU:
            # rows packed per 32: [16 values | 16 gates]
            nb = n // 32
            acc = acc.reshape(-1, nb, 2, 16)
            if bias is not None:
                acc = acc + bias.reshape(nb, 2, 16)
            v = acc[:, :, 0] * F.gelu(acc[:, :, 1])
            v = v.reshape(-1, nb * 16) * out_scale
        else:
            if bias is not None:
                acc = acc + bias
            if row_bias is not None:
                idx = torch.arange(acc.shape[0], device=acc.device) // row_div
                acc = acc + row_bias[idx]
            if act == ACT_SILU:
                acc = F.silu(acc)
            elif act == ACT_GELU:
                acc = F.gelu(acc)
            v = acc * out_scale
        if residual is not None:
            v = v + _f(residual[:v.shape[0]])
        res = self._out(v, out_f32)
        if out is not None:
            out[:v.shape[0]].copy_(res)
            return out
        return res.contiguous()

    # ------------------------------------------------------------------ attention
    def attention(self, q, k, v, *, batch, heads, lq, lk, kv_bdiv=1, out=None, accumulate=False, scale=None,
                  k2=None, v2=None, lk2=0, kv2_bdiv=1):
        scale = 64 ** -0.5 if scale is None else scale
        qf = _f(q).reshape(batch, lq, heads, 64).permute(0, 2, 1, 3)

        def one(k, v, lk, kv_bdiv):
            kvb = (batch + kv_bdiv - 1) // kv_bdiv
            kf = _f(k).reshape(kvb, lk, heads, 64).permute(0, 2, 1, 3)
            vf = _f(v).reshape(kvb, lk, heads, 64).permute(0, 2, 1, 3)
            idx = (torch.arange(batch) // kv_bdiv).tolist()
            outs = []
            for i in range(batch):
                s = (qf[i] @ kf[idx[i]].transpose(-1, -2)) * scale
                outs.append(s.softmax(-1) @ vf[idx[i]])
            return torch.stack(outs).permute(0, 2, 1, 3).reshape(batch * lq, heads * 64)
        o = one(k, v, lk, kv_bdiv)
        if k2 is not None:                     # second softmax summed in fp32, rounded once (the fused kernel's contract)
            o = o + one(k2, v2, lk2, kv2_bdiv)
        if accumulate:
            o = o + _f(out)
        o = self._out(o)
        if out is not None:
            out.copy_(o)
            return out
        return o.contiguous()

    def attention_temporal(self, qkv, *, b, t, hw, heads, scale=None):
        scale = 64 ** -0.5 if scale is None else scale
        c = heads * 64
        x = _f(qkv).reshape(b, t, hw, 3, heads, 64)
        q, k, v = (x[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))    # [b, hw, heads, t, 64]
        s = (q @ k.transpose(-1, -2)) * scale
        o = s.softmax(-1) @ v
        return self._out(o.permute(0, 3, 1, 2, 4).reshape(b * t * hw, c)).contiguous()

    # ------------------------------------------------------------------ norms
    def groupnorm(self, x, gamma, beta, *, samples, rows, eps, silu=False, part=None, prefetch=None, prefetch_linear=False):
        # `prefetch` (ABI 12, the consumer's weights): a cache hint of the HIP path -- no arithmetic, nothing to emulate
        c = x.shape[1]
        if part is not None and part.of is x and rows % part.rows == 0 and part.sums.shape[2] == c \
                and part.sums.shape[0] * part.rows == samples * rows:
            # statistics from the producer's partial sums, as tc_groupnorm_part takes them (fp64 E[x^2] - mean^2)
            self.gn_part_used += 1
            cpg = c // 32
            sums = part.sums.double().reshape(samples, rows // part.rows, 2, 32, cpg).sum(dim=(1, 4))    # [samples, 2, 32]
            cnt = rows * cpg
            mean = sums[:, 0] / cnt
            var = (sums[:, 1] / cnt - mean * mean).clamp_min(0.0)
            xf = _f(x).reshape(samples, rows, 32, cpg)
            y = (xf - mean.float()[:, None, :, None]) * (1.0 / torch.sqrt(var + eps)).float()[:, None, :, None]
            y = y.reshape(samples, rows, c) * gamma + beta
            if silu:
                y = F.silu(y)
            return self._out(y.reshape(samples * rows, c)).contiguous()
        xf = _f(x).reshape(samples, rows, c).permute(0, 2, 1)
        y = F.group_norm(xf, 32, gamma, beta, eps)
        if silu:
            y = F.silu(y)
        return self._out(y.permute(0, 2, 1).reshape(samples * rows, c)).contiguous()

    def gn_conv(self, x, gamma, beta, w, bias=None, *, samples, rows, eps, conv, silu=True, part=None, prefetch_extra=(), **kw):
        h = self.groupnorm(x, gamma, beta, samples=samples, rows=rows, eps=eps, silu=silu, part=part)
        return self.gemm(h, w, bias, conv=conv, **kw)

    def layernorm(self, x, gamma, beta, eps=1e-5, mx_for=None, prefetch=None):
        return self._out(F.layer_norm(_f(x), (x.shape[1],), gamma, beta, eps))

    def softmax_rows(self, s, n=None, causal_period=0):
        ld = s.shape[1]
        n = ld if n is None else n
        sc = s[:, :n].clone()
        if causal_period > 0:
            r = torch.arange(s.shape[0], device=s.device) % causal_period
            sc = sc.masked_fill(torch.arange(n, device=s.device)[None, :] > r[:, None], float("-inf"))
        out = torch.zeros_like(s)
        out[:, :n] = sc.softmax(-1)
        return self._out(out)

    # ------------------------------------------------------------------ layout / elementwise
    def nchw_to_rows(self, x0, x1=None, *, c_pad, scale=1.0, out=None):
        x = x0 if x1 is None else torch.cat([x0, x1], dim=1)
        b, c, t, h, w = x.shape
        rows = (x * scale).permute(0, 2, 3, 4, 1).reshape(b * t * h * w, c)
        full = torch.zeros((rows.shape[0], c_pad), dtype=torch.float32, device=x.device)
        full[:, :c] = rows
        if out is not None:
            out.copy_(full.to(BF16))
            return out
        return full.to(BF16)

    def rows_to_nchw(self, rows, *, c, b, t, h, w):
        return _f(rows[:, :c]).reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3).contiguous()

    def concat_rows(self, a, b):
        return torch.cat([a, b], dim=1).contiguous()

    def timestep_embedding(self, t, dim, ld=None):
        import math
        ld = dim if ld is None else ld
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        out = torch.zeros((t.shape[0], ld), dtype=torch.float32, device=t.device)
        out[:, :half] = torch.cos(args)
        out[:, half:2 * half] = torch.sin(args)
        return out.to(BF16)

    def silu_to_bf16(self, x):
        return F.silu(x).to(BF16)

    def repeat_rows(self, x, n):
        return x.repeat(n, 1)

    def time_mix3(self, rows, w, bias, *, b, t, h, w_):
        x = rows[:, :3].reshape(b, t, h, w_, 3).permute(0, 4, 1, 2, 3)
        return F.conv3d(x, w.reshape(3, 3, 3, 1, 1), bias, padding=(1, 0, 0)).contiguous()

    def video_to_uint8(self, video):
        v = torch.clamp(video.float(), -1., 1.)
        v = (v + 1.0) / 2.0
        return (v * 255).to(torch.uint8).permute(0, 2, 3, 4, 1).contiguous()

    def ddim_step(self, x, e_cond, e_uncond, noise, *, cfg_scale, guidance_rescale, sqrt_ac, sqrt_1m_ac,
                  sqrt_a_prev, dir_coef, sigma, x0_rescale, want_x0=True, e_uncond_img=None, cfg_img=None):
        f = lambda v: torch.tensor(v, dtype=torch.float32, device=x.device)
        v = e_cond
        if e_uncond is not None:
            if e_uncond_img is not None:
                ci = cfg_scale if cfg_img is None else cfg_img
                v = e_uncond + ci * (e_uncond_img - e_uncond) + cfg_scale * (e_cond - e_uncond_img)
            else:
                v = e_uncond + cfg_scale * (e_cond - e_uncond)
            if guidance_rescale > 0:
                dims = list(range(1, v.dim()))
                st = e_cond.double().std(dim=dims, keepdim=True)
                sc = v.double().std(dim=dims, keepdim=True)
                fac = (st / sc).float()
                v = guidance_rescale * (v * fac) + (1 - guidance_rescale) * v
        e_t = f(sqrt_ac) * v + f(sqrt_1m_ac) * x
        x0 = (f(sqrt_ac) * x - f(sqrt_1m_ac) * v) * f(x0_rescale)
        xp = f(sqrt_a_prev) * x0 + f(dir_coef) * e_t
        if noise is not None:
            xp = xp + f(sigma) * noise
        return xp, (x0 if want_x0 else None)
