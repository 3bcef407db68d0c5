"""Image-token Resampler on the HIP operators (the conditioning step before the loop).

Mirror of reference lvdm/modules/encoders/resampler.py: ImageProjModel (9-23), FeedForward (27-34),
PerceiverAttention (49-93), Resampler (96-145) -- same constructor kwargs, same parameter names
(`image_proj_model.*` of a ToonCrafter checkpoint loads strictly), same call: CLIP image tokens
(B, 257, 1280) -> (B, num_queries * video_length, output_dim) context tokens.

Same kernels as the UNet's transformer blocks: LayerNorm, bias-free Linear = tc_gemm_bf16 (exact-erf
GELU and the residual adds in its epilogue), attention = tc_attn_d64 with the 16*T learned queries
attending to [image tokens ; queries].  The K/V of the two halves of that concatenation are projected
straight into one buffer (two strided batched GEMMs), so `torch.cat((x, latents))` is never formed.
The reference scales q and k by d^-1/4 each; the attention kernel applies d^-1/2 to the fp32 scores.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU
from .common import BF16, PackedModule, f32, pack_linear


class ImageProjModel(PackedModule):
    """Projection Model (resampler.py:9-23); unused by inference_512_v1.0.yaml."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def _pack(self):
        return {"w": pack_linear(self.proj.weight), "b": f32(self.proj.bias),
                "g": f32(self.norm.weight), "beta": f32(self.norm.bias)}

    def forward(self, image_embeds):
        pk = self.pk
        rows = ops.gemm(image_embeds.reshape(-1, image_embeds.shape[-1]).to(BF16).contiguous(), pk["w"], pk["b"])
        rows = ops.layernorm(rows.reshape(-1, self.cross_attention_dim), pk["g"], pk["beta"])
        return rows.reshape(-1, self.clip_extra_context_tokens, self.cross_attention_dim).float()


def FeedForward(dim, mult=4):
    inner_dim = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim, bias=False), nn.GELU(),
                         nn.Linear(inner_dim, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("tc_attn_d64 serves head dimension 64 (the config's value)")
        self.scale = dim_head ** -0.5
        self.dim_head, self.heads = dim_head, heads
        inner_dim = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)


class Resampler(PackedModule):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        self.num_queries = num_queries
        self.video_length = video_length
        if video_length is not None:
            num_queries = num_queries * video_length
        self.dim, self.heads, self.dim_head = dim, heads, dim_head
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                              FeedForward(dim=dim, mult=ff_mult)]))

    def _pack(self):
        pk = {"lat": self.latents.detach().to(BF16).reshape(-1, self.dim).contiguous(),
              "wi": pack_linear(self.proj_in.weight), "bi": f32(self.proj_in.bias),
              "wo": pack_linear(self.proj_out.weight), "bo": f32(self.proj_out.bias),
              "og": f32(self.norm_out.weight), "ob": f32(self.norm_out.bias), "layers": []}
        for attn, ff in self.layers:
            pk["layers"].append({
                "g1": f32(attn.norm1.weight), "b1": f32(attn.norm1.bias),
                "g2": f32(attn.norm2.weight), "b2": f32(attn.norm2.bias),
                "wq": pack_linear(attn.to_q.weight), "wkv": pack_linear(attn.to_kv.weight),
                "wout": pack_linear(attn.to_out.weight),
                "fg": f32(ff[0].weight), "fb": f32(ff[0].bias),
                "w1": pack_linear(ff[1].weight), "w2": pack_linear(ff[3].weight)})
        return pk

    def forward(self, x):
        """x: (B, n_tokens, embedding_dim) -> (B, L, output_dim) fp32, L = num_queries [* video_length]."""
        pk = self.pk
        b, n1, e = x.shape
        n2 = pk["lat"].shape[0]
        inner = self.heads * self.dim_head
        xr = ops.gemm(x.reshape(b * n1, e).to(BF16).contiguous(), pk["wi"], pk["bi"])          # [b*n1, dim]
        lat = pk["lat"].repeat(b, 1)                                                          # [b*n2, dim]
        kv = torch.empty((b * (n1 + n2), 2 * inner), dtype=BF16, device=xr.device)
        for lp in pk["layers"]:
            xn = ops.layernorm(xr, lp["g1"], lp["b1"])
            ln = ops.layernorm(lat, lp["g2"], lp["b2"])
            q = ops.gemm(ln, lp["wq"])
            # K/V rows of sample i: [image tokens (n1) ; queries (n2)], like torch.cat((x, latents), dim=-2)
            ops.gemm(xn[:n1], lp["wkv"], out=kv[:n1], batch=b, stride_a=n1 * self.dim,
                     stride_c=(n1 + n2) * 2 * inner)
            ops.gemm(ln[:n2], lp["wkv"], out=kv[n1:n1 + n2], batch=b, stride_a=n2 * self.dim,
                     stride_c=(n1 + n2) * 2 * inner)
            att = ops.attention(q, kv[:, :inner], kv[:, inner:], batch=b, heads=self.heads, lq=n2, lk=n1 + n2)
            lat = ops.gemm(att, lp["wout"], residual=lat)
            h = ops.layernorm(lat, lp["fg"], lp["fb"])
            h = ops.gemm(h, lp["w1"], act=ACT_GELU)
            lat = ops.gemm(h, lp["w2"], residual=lat)
        out = ops.gemm(lat, pk["wo"], pk["bo"])
        out = ops.layernorm(out, pk["og"], pk["ob"])
        return out.reshape(b, n2, -1).float()
