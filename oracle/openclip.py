"""Oracle (test infrastructure): the two OpenCLIP ViT-H/14 towers as the reference drives them, row f2.

PINNED (round 5) to an independent third-party implementation: HuggingFace `transformers`' CLIPVisionModel / CLIPTextModel at
the ViT-H/14 geometry on seeded weights (tests/golden/make_openclip_golden.py -> tests/golden/openclip_hf.npz; replayed by
tests/test_openclip_golden_cpu.py: 5e-7 / 6e-7, and the golden separates 'penultimate' from 'last' by 0.21).  The arithmetic
the reference binds lives in a package that is absent from /root/reference and from this image: `open_clip_torch==2.22.0`
(reference requirements.txt:22; `open_clip.create_model_and_transforms("ViT-H-14", pretrained="laion2b_s32b_b79k")`,
lvdm/modules/encoders/condition.py:188,307), so the pin is against transformers under the published open_clip -> transformers
parameter renaming, not against open_clip itself.  What follows
restates its published architecture (open_clip/transformer.py @ v2.22.0: VisionTransformer, Transformer,
ResidualAttentionBlock = x + attn(ln_1(x)); x + mlp(ln_2(x)), nn.MultiheadAttention with a fused
in_proj, MLP c_fc -> GELU(erf) -> c_proj, no LayerScale for ViT-H) along the exact sequence of attribute
accesses of the reference's own call sites:
  * image: condition.py:340-372  encode_with_vision_transformer -- conv1 patchify, class token, positional
    embedding, ln_pre, ALL resblocks, NO ln_post / proj  -> (B, 257, 1280)
  * text:  condition.py:215-231  encode_with_transformer with layer="penultimate" (inference_512_v1.0.yaml:
    the last resblock is skipped), causal attn_mask, ln_final -> (B, 77, 1024)
State-dict keys are open_clip's (`visual.conv1.weight`, `visual.transformer.resblocks.N.attn.in_proj_weight`,
`token_embedding.weight`, ...), i.e. what a ToonCrafter checkpoint stores under `embedder.model.*` and
`cond_stage_model.model.*`.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

ARCH = {"ViT-H-14": dict(vision=dict(width=1280, layers=32, heads=16, patch=14, image=224, mlp=5120),
                         text=dict(width=1024, layers=24, heads=16, context=77, vocab=49408, mlp=4096))}


def resblock(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, heads: int, mask=None) -> torch.Tensor:
    """x: (B, L, D)."""
    b, l, d = x.shape
    h = F.layer_norm(x, (d,), sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = h @ sd[pre + "attn.in_proj_weight"].t() + sd[pre + "attn.in_proj_bias"]
    q, k, v = (t.view(b, l, heads, d // heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    s = (q @ k.transpose(-2, -1)) * (d // heads) ** -0.5
    if mask is not None:
        s = s + mask
    a = (s.softmax(-1) @ v).transpose(1, 2).reshape(b, l, d)
    x = x + a @ sd[pre + "attn.out_proj.weight"].t() + sd[pre + "attn.out_proj.bias"]
    h = F.layer_norm(x, (d,), sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    h = F.gelu(h @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"])
    return x + h @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]


def vision_tokens(sd: Dict[str, torch.Tensor], image: torch.Tensor, heads: int) -> torch.Tensor:
    """image: (B, 3, S, S) already resized / normalised (condition.py:323-331 is kornia, see lvdm/openclip.py).
    sd: the `visual.*` sub-dict."""
    x = F.conv2d(image, sd["conv1.weight"], stride=sd["conv1.weight"].shape[-1])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd["class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + sd["positional_embedding"]
    x = F.layer_norm(x, (x.shape[-1],), sd["ln_pre.weight"], sd["ln_pre.bias"])
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.resblocks."))
    for i in range(depth):
        x = resblock(sd, f"transformer.resblocks.{i}.", x, heads)
    return x


def text_tokens(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, heads: int, skip_last: int = 1) -> torch.Tensor:
    """tokens: (B, 77) int64.  skip_last = 1 for layer='penultimate' (condition.py:197-201,224-226)."""
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    l = x.shape[1]
    mask = torch.full((l, l), float("-inf")).triu_(1)                 # open_clip build_attention_mask
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.resblocks."))
    for i in range(depth - skip_last):
        x = resblock(sd, f"transformer.resblocks.{i}.", x, heads, mask)
    return F.layer_norm(x, (x.shape[-1],), sd["ln_final.weight"], sd["ln_final.bias"])
