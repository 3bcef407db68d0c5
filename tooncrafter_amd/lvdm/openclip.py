"""The two OpenCLIP ViT-H/14 towers on the HIP operators (conditioning before the loop).

What the reference binds: `open_clip.create_model_and_transforms("ViT-H-14", ...)` (third-party
open_clip_torch==2.22.0, absent here) driven attribute by attribute from
lvdm/modules/encoders/condition.py:215-231 (text, layer="penultimate") and :340-372 (image tokens).  This file
rebuilds those two module trees with open_clip's parameter names -- so the `cond_stage_model.model.*` and
`embedder.model.*` tensors of a ToonCrafter checkpoint load strictly -- and runs them on the same kernels as
the UNet: LayerNorm, tc_gemm_bf16 (bias / exact-erf GELU / residual epilogues), and attention as
GEMM -> row softmax -> GEMM (tc_softmax_rows with K padding and the causal text mask), because the vision
tower's head dimension is 80, not 64.  Parity: pinned to HuggingFace transformers' CLIP at the ViT-H/14 geometry
(tests/golden/make_openclip_golden.py; tests/test_openclip_golden_cpu.py for the oracle, test_gpu_models.py for this path).

Result-preserving restructurings: the patch-embedding convolution (kernel = stride = 14) is a GEMM over
unfolded patches with the positional embedding as its residual; V's bias is folded into the out-projection
bias (softmax rows sum to one: P (V + 1 b^T) W_o^T = P V W_o^T + b W_o^T).
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU
from .common import BF16, PackedModule, ceil_to, f32, pack_linear

ARCH = {"ViT-H-14": dict(embed_dim=1024,
                         vision=dict(width=1280, layers=32, heads=16, patch=14, image=224, mlp=5120),
                         text=dict(width=1024, layers=24, heads=16, context=77, vocab=49408, mlp=4096))}


class MultiheadAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d, heads):
        super().__init__()
        self.embed_dim, self.num_heads = d, heads
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads, mlp):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, mlp)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(mlp, d))]))


class Transformer(PackedModule):
    def __init__(self, width, layers, heads, mlp):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp) for _ in range(layers)])

    def _pack(self):
        d = self.width
        out = []
        for r in self.resblocks:
            w, b = r.attn.in_proj_weight.detach(), r.attn.in_proj_bias.detach()
            wo, bo = r.attn.out_proj.weight.detach(), r.attn.out_proj.bias.detach()
            out.append({
                "g1": f32(r.ln_1.weight), "b1": f32(r.ln_1.bias), "g2": f32(r.ln_2.weight), "b2": f32(r.ln_2.bias),
                "wqk": pack_linear(w[:2 * d]), "bqk": f32(b[:2 * d]), "wv": pack_linear(w[2 * d:]),
                "wo": pack_linear(wo), "bo": f32(bo.float() + wo.float() @ b[2 * d:].float()),     # V bias folded
                "w1": pack_linear(r.mlp.c_fc.weight), "bf1": f32(r.mlp.c_fc.bias),
                "w2": pack_linear(r.mlp.c_proj.weight), "bf2": f32(r.mlp.c_proj.bias)})
        return {"blocks": out}

    def run(self, x, b, l, n_blocks=None, causal=False):
        """x: bf16 rows [b*l, width] (row = sample*l + token) -> same, after the first `n_blocks` resblocks."""
        d, heads = self.width, self.heads
        dh, lp = d // heads, ceil_to(l, 8)
        blocks = self.pk["blocks"][:self.layers if n_blocks is None else n_blocks]
        dev = x.device
        vt = torch.zeros((b * d, lp), dtype=BF16, device=dev)            # V^T per sample, K-padding columns stay 0
        s = torch.empty((b * heads * l, lp), dtype=torch.float32, device=dev)
        o = torch.empty((b * l, d), dtype=BF16, device=dev)
        for pk in blocks:
            h = ops.layernorm(x, pk["g1"], pk["b1"])
            qk = ops.gemm(h, pk["wqk"], pk["bqk"])                        # [b*l, 2d]: q | k
            ops.gemm(pk["wv"], h[:l], out=vt[:d, :l], batch=b, stride_a=0, stride_w=l * d, stride_c=d * lp)
            for i in range(b):
                rows = slice(i * l, (i + 1) * l)
                ops.gemm(qk[rows, :dh], qk[rows, d:d + dh], alpha=float(dh) ** -0.5, out_f32=True,
                         out=s[i * heads * l:i * heads * l + l, :l], batch=heads, stride_a=dh, stride_w=dh,
                         stride_c=l * lp)
            p = ops.softmax_rows(s, n=l, causal_period=l if causal else 0)
            for i in range(b):
                ops.gemm(p[i * heads * l:i * heads * l + l], vt[i * d:i * d + dh], out=o[i * l:(i + 1) * l, :dh],
                         batch=heads, stride_a=l * lp, stride_w=dh * lp, stride_c=dh)
            x = ops.gemm(o, pk["wo"], pk["bo"], residual=x)
            h = ops.layernorm(x, pk["g2"], pk["b2"])
            h = ops.gemm(h, pk["w1"], pk["bf1"], act=ACT_GELU)
            x = ops.gemm(h, pk["w2"], pk["bf2"], residual=x)
        return x


class VisionTransformer(PackedModule):
    """open_clip.transformer.VisionTransformer's parameter tree (ViT-H/14: no patch dropout, no input patchnorm)."""
    input_patchnorm = False

    def __init__(self, width, layers, heads, patch, image, mlp, output_dim):
        super().__init__()
        self.patch_size, self.grid_size = (patch, patch), (image // patch, image // patch)
        self.width = width
        n_tok = self.grid_size[0] * self.grid_size[1] + 1
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(width) * width ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn(n_tok, width) * width ** -0.5)
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp)
        self.ln_post = nn.LayerNorm(width)                                # unused by the token path (condition.py:372)
        self.proj = nn.Parameter(torch.randn(width, output_dim) * width ** -0.5)

    def _pack(self):
        pos = self.positional_embedding.detach().float()
        return {"wc": pack_linear(self.conv1.weight), "pos": pos[1:].to(BF16).contiguous(),
                "cls": (self.class_embedding.detach().float() + pos[0]).to(BF16).reshape(1, -1),
                "g": f32(self.ln_pre.weight), "b": f32(self.ln_pre.bias)}

    def tokens(self, image):
        """image (B, 3, S, S), already resized + normalised -> (B, 1 + grid^2, width) fp32 (condition.py:340-372)."""
        pk = self.pk
        b, c, hh, ww = image.shape
        ps, (gh, gw) = self.patch_size[0], self.grid_size
        if (hh, ww) != (gh * ps, gw * ps):
            raise ValueError(f"vision tower expects {gh * ps}x{gw * ps} input")
        # unfold to one row per patch, K = (c, py, px) like conv1.weight.reshape(width, -1), zero-padded to 8
        pat = image.reshape(b, c, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(b * gh * gw, c * ps * ps)
        a = torch.zeros((pat.shape[0], pk["wc"].shape[1]), dtype=BF16, device=image.device)
        a[:, :pat.shape[1]] = pat.to(BF16)
        emb = ops.gemm(a, pk["wc"], residual=pk["pos"].repeat(b, 1))                    # + positional embedding
        n = gh * gw
        x = torch.empty((b, n + 1, self.width), dtype=BF16, device=image.device)       # [class token ; patches]
        x[:, 0] = pk["cls"]
        x[:, 1:] = emb.reshape(b, n, self.width)
        x = ops.layernorm(x.reshape(b * (n + 1), self.width), pk["g"], pk["b"])
        x = self.transformer.run(x, b, n + 1)
        return x.reshape(b, n + 1, self.width).float()


class CLIPText(PackedModule):
    """The text half of open_clip.model.CLIP as the reference keeps it after `del model.visual`."""

    def __init__(self, embed_dim, width, layers, heads, context, vocab, mlp):
        super().__init__()
        self.width, self.context_length = width, context
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.randn(context, width) * 0.01)
        self.transformer = Transformer(width, layers, heads, mlp)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.randn(width, embed_dim) * width ** -0.5)   # unused (no pooling)
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6593)
        self.register_buffer("attn_mask", torch.full((context, context), float("-inf")).triu_(1), persistent=False)

    def _pack(self):
        return {"emb": self.token_embedding.weight.detach().to(BF16).contiguous(),
                "pos": self.positional_embedding.detach().to(BF16).contiguous(),
                "eye": torch.eye(self.width, dtype=BF16, device=self.positional_embedding.device),
                "g": f32(self.ln_final.weight), "b": f32(self.ln_final.bias)}

    def tokens(self, tokens, skip_last=1):
        """tokens (B, 77) int64 -> (B, 77, width) fp32: condition.py:215-231 with layer='penultimate'."""
        pk = self.pk
        b, l = tokens.shape
        e = pk["emb"][tokens.reshape(-1)]                                              # gather (data movement)
        x = ops.gemm(e, pk["eye"], residual=pk["pos"][:l].repeat(b, 1))                # e + positional, rounded once
        x = self.transformer.run(x, b, l, n_blocks=self.transformer.layers - skip_last, causal=True)
        x = ops.layernorm(x, pk["g"], pk["b"])
        return x.reshape(b, l, self.width).float()


def build_text(arch="ViT-H-14"):
    a = ARCH[arch]
    return CLIPText(a["embed_dim"], **a["text"])


class _VisualHolder(nn.Module):
    """`model` of FrozenOpenCLIPImageEmbedderV2 after `del model.transformer`: only `.visual` carries weights the
    token path uses (the CLIP text parameters stay in the reference's checkpoint but are dead there too)."""

    def __init__(self, arch):
        super().__init__()
        a = ARCH[arch]
        v = a["vision"]
        self.visual = VisionTransformer(v["width"], v["layers"], v["heads"], v["patch"], v["image"], v["mlp"],
                                        a["embed_dim"])
        # The reference deletes only `model.transformer` (condition.py:309): the rest of open_clip's CLIP text half
        # stays in the module -- and therefore in every ToonCrafter checkpoint (`embedder.model.positional_embedding`,
        # `text_projection`, `logit_scale`, `token_embedding.weight`, `ln_final.*`).  Dead weights, but
        # `load_state_dict(strict=True)` (inference.py:32,44) needs a home for each of them.
        t = a["text"]
        self.positional_embedding = nn.Parameter(torch.zeros(t["context"], t["width"]))
        self.text_projection = nn.Parameter(torch.zeros(t["width"], a["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6593)
        self.token_embedding = nn.Embedding(t["vocab"], t["width"])
        self.ln_final = nn.LayerNorm(t["width"])
        self.register_buffer("attn_mask", torch.full((t["context"], t["context"]), float("-inf")).triu_(1),
                             persistent=False)
