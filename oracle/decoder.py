"""Oracle: dual-reference frame-aware VideoDecoder forward, functional fp32.

Restates (reference paths under lvdm/):
  models/autoencoder.py:112-116              AutoencoderKL.decode (post_quant_conv is
                                             SKIPPED whenever kwargs are passed)
  models/ddpm3d.py:647-679                   decode_core (z / scale_factor, timesteps kwarg)
  models/autoencoder_dualref.py:72-92        ResnetBlock.forward (GN eps 1e-6, swish)
  models/autoencoder_dualref.py:172-206      MemoryEfficientAttnBlock (1 head, d = C)
  models/autoencoder_dualref.py:270-341      MemoryEfficientCrossAttentionWrapperFusion
  models/autoencoder_dualref.py:357-368      Combiner
  models/autoencoder_dualref.py:489-527      Decoder.forward
  models/autoencoder_dualref.py:672-698      3-D ResBlock._forward (skip_t_emb)
  models/autoencoder_dualref.py:892-911      VideoResBlock.forward
  models/autoencoder_dualref.py:929-935      AE3DConv.forward
  modules/networks/openaimodel3d.py:98-106   Upsample (nearest x2 + conv3x3)
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(sd, p, x, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def _attend(q, k, v, heads, q_chunk=1024):
    """softmax(q k^T / sqrt(d)) v with heads split out of the channel axis;
    chunked over queries (level-2 reference attention is 10240 x 20480)."""
    b, lq, c = q.shape
    d = c // heads
    qh = q.reshape(b, lq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(k.shape[0], k.shape[1], heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(v.shape[0], v.shape[1], heads, d).permute(0, 2, 1, 3)
    out = torch.empty_like(qh)
    for i in range(0, lq, q_chunk):
        s = torch.matmul(qh[:, :, i:i + q_chunk], kh.transpose(-1, -2)) * (d ** -0.5)
        out[:, :, i:i + q_chunk] = torch.matmul(s.softmax(dim=-1), vh)
    return out.permute(0, 2, 1, 3).reshape(b, lq, c)


def time_stack(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """3-D ResBlock with skip_t_emb (emb term is zero), kernel (3,1,1): fp32
    GroupNorm over (T,H,W) jointly, eps 1e-5 (basics.normalization)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv3d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=(1, 0, 0))
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv3d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=(1, 0, 0))
    return x + h


def video_res_block(sd: SD, p: str, x: torch.Tensor, timesteps: int) -> torch.Tensor:
    """2-D ResnetBlock per frame, then the temporal ResBlock on (b c t h w),
    blended with alpha = sigmoid(mix_factor): alpha * x3d + (1 - alpha) * x2d."""
    h = _swish(_gn(sd, p + ".norm1", x, 1e-6))
    h = _conv(sd, p + ".conv1", h, 1)
    h = _swish(_gn(sd, p + ".norm2", h, 1e-6))
    h = _conv(sd, p + ".conv2", h, 1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, 0)
    x2d = x + h
    n, c, hh, ww = x2d.shape
    x5 = x2d.reshape(n // timesteps, timesteps, c, hh, ww).permute(0, 2, 1, 3, 4)
    x3d = time_stack(sd, p + ".time_stack", x5)
    alpha = torch.sigmoid(sd[p + ".mix_factor"])
    y = alpha * x3d + (1.0 - alpha) * x5
    return y.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def mid_attention(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Single-head self attention with d = C over the H*W tokens of each frame."""
    n, c, hh, ww = x.shape
    h = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (_conv(sd, f"{p}.{nm}", h, 0).reshape(n, c, hh * ww).transpose(1, 2) for nm in "qkv")
    o = _attend(q, k, v, heads=1)
    o = o.transpose(1, 2).reshape(n, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", o, 0)


def ref_fusion(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """Queries: GN(x) tokens of every frame.  Keys/values: tokens of the TWO
    reference frames (no norm on the context), concatenated along the key axis
    [first-frame HW tokens | last-frame HW tokens]; the same K/V serve every one
    of the bt//b frames of a clip."""
    bt, c, hh, ww = x.shape
    q = F.linear(_gn(sd, p + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(bt, hh * ww, c),
                 sd[p + ".to_q.weight"])
    b, cc, l, hc, wc = context.shape
    ctx = context.permute(0, 2, 3, 4, 1).reshape(b, l * hc * wc, cc)      # [b, (l hw), c]
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    rep = bt // b
    outs = []
    for i in range(b):      # identical K/V for every frame of clip i: never materialise the repeat
        qi = q[i * rep:(i + 1) * rep]
        outs.append(_attend(qi, k[i:i + 1].expand(rep, -1, -1), v[i:i + 1].expand(rep, -1, -1), heads))
    o = torch.cat(outs, 0)
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.reshape(bt, hh, ww, c).permute(0, 3, 1, 2)


def combiner(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """1x1 conv of the two reference frames, added to the first and the last frame
    of each clip (two separate adds: with one frame per clip both land on it)."""
    b, c, l, hh, ww = context.shape
    bt = x.shape[0]
    t = bt // b
    ctx = context.permute(0, 2, 1, 3, 4).reshape(b * l, c, hh, ww)
    ctx = _conv(sd, p + ".conv", ctx, 0).reshape(b, l, c, hh, ww)
    x5 = x.reshape(b, t, c, hh, ww).clone()
    x5[:, 0] = x5[:, 0] + ctx[:, 0]
    x5[:, -1] = x5[:, -1] + ctx[:, 1]
    return x5.reshape(bt, c, hh, ww)


def decoder_forward(sd: SD, z: torch.Tensor, ref_context: List[torch.Tensor], timesteps: int,
                    num_levels: int = 4, num_res_blocks: int = 2, probe=None) -> torch.Tensor:
    """Decoder.forward with kwargs={'timesteps': T}.  `probe(name, tensor)`, when given, sees the activation
    after the mid block and after every level's reference fusion (test instrumentation only).
    z: (B*T, zc, h, w) already
    divided by scale_factor; ref_context: five (B, C, 2, H, W) tensors indexed by
    level (0 = full resolution) plus the final one.  -> (B*T, 3, 8h, 8w)."""
    h = _conv(sd, "conv_in", z, 1)
    h = video_res_block(sd, "mid.block_1", h, timesteps)
    h = mid_attention(sd, "mid.attn_1", h)
    h = video_res_block(sd, "mid.block_2", h, timesteps)
    if probe is not None:
        probe("mid", h)                                   # parity tests compare per stage: (B*T, C, H, W)
    for lvl in reversed(range(num_levels)):
        for ib in range(num_res_blocks + 1):
            h = video_res_block(sd, f"up.{lvl}.block.{ib}", h, timesteps)
        ar = f"attn_refinement.{lvl}"
        if (ar + ".to_q.weight") in sd:
            h = ref_fusion(sd, ar, h, ref_context[lvl])
        else:
            h = combiner(sd, ar, h, ref_context[lvl])
        if probe is not None:
            probe(f"level{lvl}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv(sd, f"up.{lvl}.upsample.conv", h, 1)
    h = _swish(_gn(sd, "norm_out", h, 1e-6))
    h = combiner(sd, f"attn_refinement.{num_levels}", h, ref_context[-1])
    h = _conv(sd, "conv_out", h, 1)
    n, c, hh, ww = h.shape
    h5 = h.reshape(n // timesteps, timesteps, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["conv_out.time_mix_conv.weight"], sd["conv_out.time_mix_conv.bias"],
                  padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def decode_first_stage(sd: SD, z: torch.Tensor, ref_context: List[torch.Tensor],
                       scale_factor: float = 0.18215, **kw) -> torch.Tensor:
    """ddpm3d.py:647-679 for a (B, C, T, h, w) latent with B == 1 per chunk, as the
    scripts enforce (inference.py:296): one decoder call over all T frames."""
    b, c, t, hh, ww = z.shape
    zz = (1.0 / scale_factor) * z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    out = decoder_forward(sd, zz, ref_context, timesteps=t, **kw)
    return out.reshape(b, t, out.shape[1], out.shape[2], out.shape[3]).permute(0, 2, 1, 3, 4)
