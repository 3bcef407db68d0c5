"""Transformer blocks of the UNet on the HIP operators.

Mirrors the interface of reference lvdm/modules/attention.py (class names,
constructor kwargs, parameter names -> identical state-dict keys):
CrossAttention (42-209), BasicTransformerBlock (212-246), SpatialTransformer
(249-310), TemporalTransformer (313-412), GEGLU/FeedForward (415-442).

Differences in HOW (not what) is computed:
  * q/k/v projections of a self-attention run as one fused [3C, C] GEMM;
  * every `+ x` residual and bias is folded into the epilogue of the GEMM that
    produces the branch; GEGLU is the epilogue of the first FF GEMM;
  * cross-attention K/V depend only on the conditioning, which is constant over
    the DDIM loop: they are projected once per context (ContextCache) instead of
    once per UNet call, and the text keys are shared by the T frames of a clip
    (kv_bdiv) instead of being repeat_interleave'd;
  * activations stay channels-last, so spatial tokens, temporal tokens and conv
    pixels are the same rows and no rearrange/permute is ever materialised.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GEGLU
from .common import Act, CfgShare, PackedModule, SourceKey, f32, fold_layernorm, pack_geglu, pack_linear


class ContextCache:
    """Cross-attention conditioning prepared once per sampling run.

    `context`: (B, L, Cc) fp32.  With L == 77 + 16*T the tail is per-frame image
    tokens (reference openaimodel3d.py:556-560); otherwise both parts are shared
    by all frames of a clip.  The bf16 token rows and every attention block's
    projected K/V live in buffers that are REFRESHED IN PLACE when the conditioning
    values change, so a captured hipGraph of the UNet keeps pointing at valid data."""

    def __init__(self, context: torch.Tensor, t: int, text_len: int = 77):
        b, l, cc = context.shape
        self.b, self.t, self.cc = b, t, cc
        self.shape = tuple(context.shape)
        self.text_len = min(text_len, l)
        li = l - self.text_len
        self.img_rows = None
        self.img_len = 0
        self.img_per_frame = False
        if li > 0:
            if l == 77 + t * 16:
                self.img_per_frame = True
                self.img_len = 16
            else:
                self.img_len = li
        self.text_rows = torch.empty((b * self.text_len, cc), dtype=torch.bfloat16, device=context.device)
        if li > 0:
            self.img_rows = torch.empty((b * li, cc), dtype=torch.bfloat16, device=context.device)
        self.kv = {}          # id(module) -> (module, kv_text, kv_img)
        self.key = None
        self.epoch = -1       # PackedModule.graph_epoch() the K/V were projected under (new weights = stale K/V)
        self.refresh(context)

    def is_current(self, context: torch.Tensor) -> bool:
        """True when the cache was filled from this very tensor object at its current version."""
        return self.key is not None and self.epoch == PackedModule.graph_epoch() and self.key.same([context])

    def invalidate(self):
        self.key = None

    def matches(self, context: torch.Tensor, t: int) -> bool:
        return self.shape == tuple(context.shape) and self.t == t and self.text_rows.device == context.device

    def branch(self, k: int, b: int) -> "CtxBranch":
        """Samples [k b, (k + 1) b) of this conditioning: what pass k of a guided step attends to (CfgShare branches)."""
        return CtxBranch(self, k, b)

    def refresh(self, context: torch.Tensor):
        """(Re)load the token rows and recompute every cached K/V projection in place."""
        ctx = context.detach()
        b, cc = self.b, self.cc
        self.text_rows.copy_(ctx[:, :self.text_len].reshape(b * self.text_len, cc))
        if self.img_rows is not None:
            self.img_rows.copy_(ctx[:, self.text_len:].reshape(-1, cc))
        for module, kv_text, kv_img in self.kv.values():
            module.project_context(self, kv_text, kv_img)
        self.key = SourceKey([context])
        self.epoch = PackedModule.graph_epoch()


class CtxBranch:
    """A batch slice of a ContextCache (same token geometry; K/V are row slices of the parent's projections)."""

    def __init__(self, parent: ContextCache, k: int, b: int):
        if (k + 1) * b > parent.b:
            raise ValueError(f"context holds {parent.b} samples: no slice {k} of {b}")
        self.parent, self.k, self.b, self.t = parent, k, b, parent.t
        self.text_len, self.img_len, self.img_per_frame = parent.text_len, parent.img_len, parent.img_per_frame
        self.img_rows = parent.img_rows

    def rows_of(self, kv: torch.Tensor) -> torch.Tensor:
        per = kv.shape[0] // self.parent.b                      # token rows per sample (77; 16 T or the shared image tokens)
        return kv[self.k * self.b * per:(self.k + 1) * self.b * per]


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(PackedModule):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        if not glu:
            raise NotImplementedError("the ToonCrafter config only uses the gated (GEGLU) feed-forward")
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def _pack(self):
        w1, b1 = pack_geglu(self.net[0].proj.weight, self.net[0].proj.bias)
        return {"w1": w1, "b1": b1, "w2": pack_linear(self.net[2].weight), "b2": f32(self.net[2].bias)}

    def ln_consumer(self):
        """(packed weight, gated) of the GEMM that reads the LayerNorm in front of this block."""
        return self.pk["w1"], True

    def forward(self, x_norm, residual, ln=None):
        """`ln` = (folded packed weight, folded bias, eps): `x_norm` is then the RAW rows and the LayerNorm runs as the
        prologue of the first GEMM (BasicTransformerBlock._pre)."""
        pk = self.pk
        if ln is not None and x_norm is residual:
            # level 0 (C = 320, hidden 1280): LayerNorm, both products, GEGLU and the residual as ONE launch -- the
            # [rows, 1280] hidden tensor (210 MB per block at the BASELINE shape) never reaches HBM (csrc/ff_fused.hip)
            probe = getattr(ops.backend(), "ff_fused_eligible", None)
            if probe is not None and probe(x_norm.shape[0], x_norm.shape[1], pk["w2"].shape[1], ldx=x_norm.stride(0)):
                return ops.ff_geglu_fused(x_norm, ln[0], ln[1], pk["w2"], pk["b2"], ln_eps=ln[2])
        if ln is not None:
            g = ops.gemm(x_norm, ln[0], ln[1], act=ACT_GEGLU, a_norm_eps=ln[2])
        else:
            g = ops.gemm(x_norm, pk["w1"], pk["b1"], act=ACT_GEGLU)
        return ops.gemm(g, pk["w2"], pk["b2"], residual=residual)


class CrossAttention(PackedModule):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.,
                 relative_position=False, temporal_length=None, video_length=None,
                 image_cross_attention=False, image_cross_attention_scale=1.0,
                 image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("HIP attention kernels are specialised for head dim 64")
        if relative_position:
            raise NotImplementedError("relative position is unused by inference_512_v1.0.yaml")
        if image_cross_attention_scale_learnable:
            raise NotImplementedError("learnable image cross-attention scale is unused by the config")
        inner = dim_head * heads
        self.is_self = context_dim is None
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.temporal_length = temporal_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale = image_cross_attention_scale
        self.text_context_len = text_context_len
        if image_cross_attention:
            self.to_k_ip = nn.Linear(context_dim, inner, bias=False)
            self.to_v_ip = nn.Linear(context_dim, inner, bias=False)

    def _pack(self):
        pk = {"wo": pack_linear(self.to_out[0].weight), "bo": f32(self.to_out[0].bias)}
        if self.is_self:
            pk["wqkv"] = pack_linear(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0))
        else:
            pk["wq"] = pack_linear(self.to_q.weight)
            pk["wkv"] = pack_linear(torch.cat([self.to_k.weight, self.to_v.weight], 0))
            if self.image_cross_attention:
                pk["wkv_ip"] = pack_linear(torch.cat([self.to_k_ip.weight, self.to_v_ip.weight], 0))
        return pk

    # -- self attention over the H*W tokens of each frame
    def forward_spatial_self(self, x_norm, residual, act: Act, ln=None):
        pk = self.pk
        c = self.heads * 64
        qkv = ops.gemm(x_norm, pk["wqkv"]) if ln is None else ops.gemm(x_norm, ln[0], ln[1], a_norm_eps=ln[2])
        a = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], batch=act.frames, heads=self.heads,
                          lq=act.hw, lk=act.hw, scale=self.scale)
        return ops.gemm(a, pk["wo"], pk["bo"], residual=residual)

    # -- self attention over the T frames at each pixel
    def forward_temporal_self(self, x_norm, residual, act: Act, ln=None):
        pk = self.pk
        # levels 1-3: the fused projection and the 16 x 16 attentions as ONE launch (csrc/qkv_attn.hip, ABI 13) -- the
        # [rows, 3C] tensor between them never reaches HBM; the library's own rule decides (16 frames, hw % 8 == 0)
        fused = getattr(ops.backend(), "temporal_qkv_attn_eligible", None) if ln is None and torch.is_tensor(x_norm) else None
        if fused is not None and fused(b=act.b, t=act.t, hw=act.hw, c=x_norm.shape[1], heads=self.heads, ldx=x_norm.stride(0)):
            a = ops.temporal_qkv_attn(x_norm, pk["wqkv"], None, b=act.b, t=act.t, hw=act.hw, heads=self.heads, scale=self.scale)
            return ops.gemm(a, pk["wo"], pk["bo"], residual=residual)
        qkv = ops.gemm(x_norm, pk["wqkv"]) if ln is None else ops.gemm(x_norm, ln[0], ln[1], a_norm_eps=ln[2])
        a = ops.attention_temporal(qkv, b=act.b, t=act.t, hw=act.hw, heads=self.heads, scale=self.scale)
        return ops.gemm(a, pk["wo"], pk["bo"], residual=residual)

    # -- text + image cross attention: two softmaxes, summed
    def project_context(self, ctx: ContextCache, kv_text=None, kv_img=None):
        pk = self.pk
        kv_text = ops.gemm(ctx.text_rows, pk["wkv"], out=kv_text)
        if self.image_cross_attention and ctx.img_rows is not None:
            kv_img = ops.gemm(ctx.img_rows, pk["wkv_ip"], out=kv_img)
        return kv_text, kv_img

    def context_kv(self, ctx):
        root = getattr(ctx, "parent", ctx)                      # a CtxBranch reads row slices of its parent's K/V
        hit = root.kv.get(id(self))
        if hit is None:
            kv_text, kv_img = self.project_context(root)
            hit = root.kv[id(self)] = (self, kv_text, kv_img)
        if root is ctx:
            return hit[1], hit[2]
        return ctx.rows_of(hit[1]), None if hit[2] is None else ctx.rows_of(hit[2])

    def forward_cross(self, x_norm, residual, act: Act, ctx: ContextCache, ln=None):
        pk = self.pk
        c = self.heads * 64
        kv_text, kv_img = self.context_kv(ctx)
        if kv_img is not None and self.image_cross_attention_scale != 1.0:
            raise NotImplementedError("image_cross_attention_scale != 1.0")
        q = ops.gemm(x_norm, pk["wq"]) if ln is None else ops.gemm(x_norm, ln[0], ln[1], a_norm_eps=ln[2])
        if kv_img is not None:
            # text and image softmaxes in ONE launch (attention.py:153-207 runs two attentions and adds them): Q is read
            # once and the sum is formed in fp32 registers -- no bf16 round trip of the first result through HBM
            a = ops.attention(q, kv_text[:, :c], kv_text[:, c:], batch=act.frames, heads=self.heads, lq=act.hw,
                              lk=ctx.text_len, kv_bdiv=act.t, scale=self.scale,
                              k2=kv_img[:, :c], v2=kv_img[:, c:], lk2=ctx.img_len,
                              kv2_bdiv=1 if ctx.img_per_frame else act.t)
        else:
            a = ops.attention(q, kv_text[:, :c], kv_text[:, c:], batch=act.frames, heads=self.heads, lq=act.hw,
                              lk=ctx.text_len, kv_bdiv=act.t, scale=self.scale)
        return ops.gemm(a, pk["wo"], pk["bo"], residual=residual)


class BasicTransformerBlock(PackedModule):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attention_cls=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is unused by the config")
        attn_cls = CrossAttention if attention_cls is None else attention_cls
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                              dropout=dropout, video_length=video_length,
                              image_cross_attention=image_cross_attention,
                              image_cross_attention_scale=image_cross_attention_scale,
                              image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                              text_context_len=text_context_len)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    LN_EPS = 1e-5

    def _pack(self):
        return {f"g{i}": f32(getattr(self, f"norm{i}").weight) for i in (1, 2, 3)} | \
               {f"b{i}": f32(getattr(self, f"norm{i}").bias) for i in (1, 2, 3)}

    def _ln(self, x, i, consumer=None, gated=False, prefetch=None):
        """`consumer`: the packed weight of the single linear GEMM that reads the normalised rows (fused qkv, GEGLU
        projection) -- lets the fp8 route have LayerNorm emit MXFP8 directly; ignored otherwise.  `prefetch`: packed
        weights of the GEMMs behind this norm (ABI 12: the norm's launch streams them ahead where that pays)."""
        mx_for = None if consumer is None else (consumer.shape[0], consumer.shape[0] // 2 if gated else consumer.shape[0])
        return ops.layernorm(x, self.pk[f"g{i}"], self.pk[f"b{i}"], self.LN_EPS, mx_for=mx_for, prefetch=prefetch)

    def _folded(self, i, kind):
        """Consumer weights of norm<i> with the LayerNorm's affine half folded in (common.fold_layernorm), packed like
        the plain ones.  Cached in the CONSUMER's packed dict (attn.pk / ff.pk), so reloading that child alone drops
        them with the rest of its packed tensors, and stamped with the norm's parameter versions, so reloading the
        norm alone rebuilds them (the block-level cache saw neither)."""
        norm = getattr(self, f"norm{i}")
        child = self.ff if kind == "ff" else (self.attn1 if i == 1 else self.attn2)
        cpk = child.pk
        key = f"ln_fold_{kind}"
        stamp = (id(norm.weight), norm.weight._version, id(norm.bias), norm.bias._version)
        ent = cpk.get(key)
        if ent is None or ent[0] != stamp:
            with torch.no_grad():
                if kind == "ff":
                    w, b = fold_layernorm(self.ff.net[0].proj.weight, self.ff.net[0].proj.bias, norm.weight, norm.bias)
                    folded = pack_geglu(w, b)
                else:
                    raw = torch.cat([child.to_q.weight, child.to_k.weight, child.to_v.weight], 0) if kind == "qkv" \
                        else child.to_q.weight
                    w, b = fold_layernorm(raw, None, norm.weight, norm.bias)
                    folded = (pack_linear(w), b)
            ent = cpk[key] = (stamp, folded)
        return (*ent[1], self.LN_EPS)

    def _pre(self, x, i, consumer, kind, then=None):
        """Input of the GEMM behind norm<i>: (LayerNorm(x), None), or -- when that GEMM can normalise its A rows itself
        (K = 320 at level 0: ops.gemm_ln_eligible, the library's own rule) -- (x, folded consumer weights): the
        LayerNorm launch and its HBM round trip (reference attention.py:242-246 runs it as its own kernel) disappear.
        `then`: the packed weight of the GEMM that follows the consumer (output projection, second feed-forward layer):
        prefetched with the consumer's by the norm's launch (ABI 12)."""
        gated = kind == "ff"
        if gated:   # the one-launch feed-forward (csrc/ff_fused.hip) normalises its rows itself: raw rows + folded weights
            fused = getattr(ops.backend(), "ff_fused_eligible", None)
            if fused is not None and fused(x.shape[0], x.shape[1], self.ff.pk["w2"].shape[1], ldx=x.stride(0)):
                return x, self._folded(i, kind)
        probe = getattr(ops.backend(), "gemm_ln_eligible", None)
        if probe is not None and probe(x.shape[0], consumer.shape[0], consumer.shape[1], geglu=gated, lda=x.stride(0)):
            return x, self._folded(i, kind)
        return self._ln(x, i, consumer if kind != "q" else None, gated, prefetch=[consumer, then]), None

    def forward_spatial(self, x, act: Act, ctx: ContextCache, share: CfgShare = None):
        """`share` (first block of the UNet's first spatial transformer under batched guidance): x / act arrive at the
        single-copy batch; the guided passes part ways at the cross-attention, so the rows are repeated in front of it
        and (x, expanded act) is returned."""
        def self_attn(x=x):
            h, ln = self._pre(x, 1, self.attn1.pk["wqkv"], "qkv", then=self.attn1.pk["wo"])
            return self.attn1.forward_spatial_self(h, x, act, ln=ln)
        x = self_attn() if share is None else share.cached(("attn1", id(self)), self_attn)
        if share is not None:
            if not share.branches:
                x, act = ops.repeat_rows(x, share.n), share.expand(act)
            share.split(x)                                               # branches: every pass goes on alone, at batch b
        h, ln = self._pre(x, 2, self.attn2.pk["wq"], "q", then=self.attn2.pk["wo"])
        x = self.attn2.forward_cross(h, x, act, ctx, ln=ln)
        h, ln = self._pre(x, 3, self.ff.pk["w1"], "ff", then=self.ff.pk["w2"])
        out = self.ff(h, x, ln=ln)
        return out if share is None else (out, act)

    def _temporal_attn(self, x, i, attn, act: Act):
        """x + attn(norm<i>(x)) over the frames of every pixel: LayerNorm, then the q / k / v projection and the 16 x 16
        attentions as ONE launch (csrc/qkv_attn.hip via CrossAttention.forward_temporal_self), then the output projection
        with the residual.  Where the library offers it (TC_TB_FUSED=1, C = 320: csrc/tb_fused.hip) all of that is one launch
        -- the default of rounds 4-5 at level 0, behind the chain above since round 6 (+0.4 ... +0.6 % per forward)."""
        fused = getattr(ops.backend(), "temporal_attn_fused_eligible", None)
        if fused is not None and fused(b=act.b, t=act.t, hw=act.hw, c=x.shape[1], heads=attn.heads, ldx=x.stride(0)):
            w, bias, eps = self._folded(i, "qkv")
            return ops.temporal_attn_fused(x, w, bias, attn.pk["wo"], attn.pk["bo"], b=act.b, t=act.t, hw=act.hw,
                                           heads=attn.heads, ln_eps=eps, scale=attn.scale)
        h, ln = self._pre(x, i, attn.pk["wqkv"], "qkv", then=attn.pk["wo"])
        return attn.forward_temporal_self(h, x, act, ln=ln)

    def forward_temporal(self, x, act: Act):
        x = self._temporal_attn(x, 1, self.attn1, act)
        x = self._temporal_attn(x, 2, self.attn2, act)                   # context=None -> self attention again
        h, ln = self._pre(x, 3, self.ff.pk["w1"], "ff", then=self.ff.pk["w2"])
        return self.ff(h, x, ln=ln)


class SpatialTransformer(PackedModule):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None,
                 use_checkpoint=True, disable_self_attn=False, use_linear=False, video_length=None,
                 image_cross_attention=False, image_cross_attention_scale_learnable=False):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                  disable_self_attn=disable_self_attn, checkpoint=use_checkpoint,
                                  video_length=video_length, image_cross_attention=image_cross_attention,
                                  image_cross_attention_scale_learnable=image_cross_attention_scale_learnable)
            for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear else nn.Conv2d(inner, in_channels, 1)

    def _pack(self):
        return {"gn_g": f32(self.norm.weight), "gn_b": f32(self.norm.bias),
                "wi": pack_linear(self.proj_in.weight), "bi": f32(self.proj_in.bias),
                "wo": pack_linear(self.proj_out.weight), "bo": f32(self.proj_out.bias)}

    def forward(self, act: Act, ctx: ContextCache, share: CfgShare = None) -> Act:
        pk = self.pk

        def proj_in():
            h = ops.groupnorm(act.rows, pk["gn_g"], pk["gn_b"], samples=act.frames, rows=act.hw, eps=1e-6, prefetch=[pk["wi"]], prefetch_linear=True)
            return ops.gemm(h, pk["wi"], pk["bi"])
        h = proj_in() if share is None else share.cached(("proj_in", id(self)), proj_in)
        for blk in self.transformer_blocks:
            if share is not None and not share.done:
                h, act = blk.forward_spatial(h, act, ctx, share)        # act: now the n-fold batch (proj_out's residual)
            else:
                h = blk.forward_spatial(h, act, ctx)
        return act.like(ops.gemm(h, pk["wo"], pk["bo"], residual=act.rows))


class TemporalTransformer(PackedModule):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None,
                 use_checkpoint=True, use_linear=False, only_self_att=True, causal_attention=False,
                 causal_block_size=1, relative_position=False, temporal_length=None):
        super().__init__()
        if not only_self_att or causal_attention or relative_position:
            raise NotImplementedError("only the non-causal self-attention temporal transformer of the config")
        if temporal_length is not None and temporal_length > 16:
            raise NotImplementedError("temporal attention kernel handles up to 16 frames")
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=None,
                                  checkpoint=use_checkpoint) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear else nn.Conv1d(inner, in_channels, 1)

    _pack = SpatialTransformer._pack

    def forward(self, act: Act) -> Act:
        pk = self.pk
        h = ops.groupnorm(act.rows, pk["gn_g"], pk["gn_b"], samples=act.b, rows=act.t * act.hw, eps=1e-6, prefetch=[pk["wi"]], prefetch_linear=True)
        h = ops.gemm(h, pk["wi"], pk["bi"])
        for blk in self.transformer_blocks:
            h = blk.forward_temporal(h, act)
        return act.like(ops.gemm(h, pk["wo"], pk["bo"], residual=act.rows))
