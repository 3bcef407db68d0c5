"""Oracle: spatio-temporal UNet forward, functional fp32 PyTorch.

Driven by a state dict carrying the reference's key names
(`input_blocks.1.0.in_layers.0.weight`, `...temopral_conv.conv1.2.weight`, ...)
plus a small config dict.  Block structure is recovered from which keys exist,
so one function serves the full 320-channel model and the tiny golden configs.

Restates (reference paths under lvdm/):
  modules/networks/openaimodel3d.py:36-48   TimestepEmbedSequential dispatch
  modules/networks/openaimodel3d.py:210-236 ResBlock._forward
  modules/networks/openaimodel3d.py:272-279 TemporalConvBlock.forward
  modules/networks/openaimodel3d.py:548-603 UNetModel.forward
  modules/attention.py:81-144               CrossAttention.forward
  modules/attention.py:242-246              BasicTransformerBlock._forward
  modules/attention.py:294-310              SpatialTransformer.forward
  modules/attention.py:365-412              TemporalTransformer.forward
  modules/attention.py:415-442              GEGLU / FeedForward
  models/utils_diffusion.py:8-28            timestep_embedding
  basics.py:76-87                           GroupNormSpecific (fp32 statistics)
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """utils_diffusion.py:19-23: [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(P) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[p + ".weight"]
    if w.dim() == 3:  # Conv1d k=1 used by init_attn (attention.py:332-334)
        w = w[:, :, 0]
    return F.linear(x, w, sd.get(p + ".bias"))


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """GroupNorm(32) with statistics over every non-(N,C) axis: per frame for
    4-D input, clip-wide (T,H,W jointly) for 5-D input."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, n, c = t.shape
    return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3)


def _sdpa(q, k, v, heads, chunk=64):
    """softmax(q k^T / sqrt(d)) v, heads folded into batch, chunked over batch so
    the L=2560 score matrix stays small.  attention.py:101-120 (scale = d^-0.5)."""
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    outs = []
    for i in range(0, q.shape[0], chunk):
        qs, ks, vs = q[i:i + chunk], k[i:i + chunk], v[i:i + chunk]
        sim = torch.matmul(qs, ks.transpose(-1, -2)) * (qs.shape[-1] ** -0.5)
        outs.append(torch.matmul(sim.softmax(dim=-1), vs))
    o = torch.cat(outs, 0)
    b, h, n, d = o.shape
    return o.permute(0, 2, 1, 3).reshape(b, n, h * d)


def cross_attention(sd: SD, p: str, x: torch.Tensor, context, heads: int,
                    text_len: int = 77) -> torch.Tensor:
    """attention.py:81-144.  context None -> self attention.  With context and
    image_cross_attention (keys to_k_ip present): two independent softmaxes over
    the text tokens [:77] and the per-frame image tokens [77:], summed with
    scale 1.0 (no learnable alpha in this config)."""
    q = _lin(sd, p + ".to_q", x)
    if context is None:
        out = _sdpa(q, _lin(sd, p + ".to_k", x), _lin(sd, p + ".to_v", x), heads)
    else:
        ctx_t = context[:, :text_len]
        out = _sdpa(q, _lin(sd, p + ".to_k", ctx_t), _lin(sd, p + ".to_v", ctx_t), heads)
        if (p + ".to_k_ip.weight") in sd:
            ctx_i = context[:, text_len:]
            out_ip = _sdpa(q, _lin(sd, p + ".to_k_ip", ctx_i), _lin(sd, p + ".to_v_ip", ctx_i), heads)
            if (p + ".alpha") in sd:
                out = out + out_ip * (torch.tanh(sd[p + ".alpha"]) + 1)
            else:
                out = out + 1.0 * out_ip
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """attention.py:420-442: GEGLU (x * gelu(gate), exact erf GELU) then Linear."""
    xg = _lin(sd, p + ".net.0.proj", x)
    a, gate = xg.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def transformer_block(sd: SD, p: str, x: torch.Tensor, context, heads: int) -> torch.Tensor:
    """attention.py:242-246.  attn1 is always self attention (disable_self_attn
    False); attn2 is cross attention for spatial blocks and, because the temporal
    transformer passes context=None, a second self attention for temporal ones."""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, context, d_head: int) -> torch.Tensor:
    """attention.py:294-310 with use_linear=True.  x: (B*T, C, H, W)."""
    n, c, h, w = x.shape
    y = _gn(sd, p + ".norm", x, 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(n, h * w, c)
    y = _lin(sd, p + ".proj_in", y)
    heads = y.shape[-1] // d_head
    y = transformer_block(sd, p + ".transformer_blocks.0", y, context, heads)
    y = _lin(sd, p + ".proj_out", y)
    y = y.reshape(n, h, w, c).permute(0, 3, 1, 2)
    return y + x


def temporal_transformer(sd: SD, p: str, x: torch.Tensor, d_head: int) -> torch.Tensor:
    """attention.py:365-412, only_self_att, non-causal, no relative position.
    x: (B, C, T, H, W); tokens are the T frames at each pixel."""
    b, c, t, h, w = x.shape
    y = _gn(sd, p + ".norm", x, 1e-6)
    y = y.permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)
    y = _lin(sd, p + ".proj_in", y)
    heads = y.shape[-1] // d_head
    y = transformer_block(sd, p + ".transformer_blocks.0", y, None, heads)
    y = _lin(sd, p + ".proj_out", y)
    y = y.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2)
    return y + x


def temporal_conv_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """openaimodel3d.py:272-279: four [GroupNorm over (T,H,W) jointly, SiLU,
    Conv3d (3,1,1) pad (1,0,0)], residual.  conv1 is Sequential(GN,SiLU,Conv)
    -> conv index 2; conv2..4 have a Dropout in between -> conv index 3."""
    y = x
    for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
        q = f"{p}.conv{i}"
        y = F.silu(_gn(sd, q + ".0", y, 1e-5))
        y = F.conv3d(y, sd[f"{q}.{ci}.weight"], sd[f"{q}.{ci}.bias"], padding=(1, 0, 0))
    return x + y


def res_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, batch: int) -> torch.Tensor:
    """openaimodel3d.py:210-236 (no up/down, no scale-shift).  x: (B*T, C, H, W),
    emb: (B*T, E)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        ws = sd[p + ".skip_connection.weight"]
        x = F.conv2d(x, ws, sd[p + ".skip_connection.bias"], padding=ws.shape[-1] // 2)
    h = x + h
    if (p + ".temopral_conv.conv1.0.weight") in sd:
        n, c, hh, ww = h.shape
        h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
    return h


def _run_sequential(sd: SD, p: str, h, emb, context, batch: int, d_head: int):
    """openaimodel3d.py:36-48: walk children `p.0`, `p.1`, ... and dispatch on
    what kind of block the keys describe."""
    i = 0
    while True:
        q = f"{p}.{i}"
        if (q + ".in_layers.0.weight") in sd:                      # ResBlock
            h = res_block(sd, q, h, emb, batch)
        elif (q + ".transformer_blocks.0.attn2.to_k_ip.weight") in sd or \
                ((q + ".transformer_blocks.0.norm1.weight") in sd and
                 sd[q + ".transformer_blocks.0.attn2.to_k.weight"].shape[1] != sd[q + ".transformer_blocks.0.attn2.to_q.weight"].shape[1]):
            h = spatial_transformer(sd, q, h, context, d_head)     # SpatialTransformer
        elif (q + ".transformer_blocks.0.norm1.weight") in sd:     # TemporalTransformer
            n, c, hh, ww = h.shape
            h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, q, h5, d_head)
            h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
        elif (q + ".op.weight") in sd:                             # Downsample
            h = F.conv2d(h, sd[q + ".op.weight"], sd[q + ".op.bias"], stride=2, padding=1)
        elif (q + ".conv.weight") in sd:                           # Upsample
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[q + ".conv.weight"], sd[q + ".conv.bias"], padding=1)
        elif (q + ".weight") in sd:                                # bare conv (input_blocks.0.0)
            h = F.conv2d(h, sd[q + ".weight"], sd[q + ".bias"], padding=1)
        else:
            break
        i += 1
    return h


def unet_forward(sd: SD, cfg: dict, x: torch.Tensor, timesteps: torch.Tensor,
                 context: torch.Tensor, fs: torch.Tensor | None = None) -> torch.Tensor:
    """openaimodel3d.py:548-603.  x: (B, Cin, T, H, W) fp32; context: (B, 77+16T, Cc)
    -> (B, Cout, T, H, W)."""
    b, _, t, _, _ = x.shape
    mc = cfg["model_channels"]
    d_head = cfg["num_head_channels"]
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, mc))))
    l_ctx = context.shape[1]
    if l_ctx == 77 + t * 16:                       # openaimodel3d.py:556-560 (hard-coded split)
        ctx_text = context[:, :77].repeat_interleave(t, dim=0)
        ctx_img = context[:, 77:].reshape(b * t, 16, context.shape[-1])
        ctx = torch.cat([ctx_text, ctx_img], dim=1)
    else:
        ctx = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
    if "fps_embedding.0.weight" in sd:
        if fs is None:
            fs = torch.full((b,), cfg.get("default_fs", 4), dtype=torch.long, device=x.device)
        fe = _lin(sd, "fps_embedding.2", F.silu(_lin(sd, "fps_embedding.0", timestep_embedding(fs, mc))))
        emb = emb + fe.repeat_interleave(t, dim=0)
    h = h.float()
    hs = []
    i = 0
    while f"input_blocks.{i}.0.weight" in sd or f"input_blocks.{i}.0.in_layers.0.weight" in sd \
            or f"input_blocks.{i}.0.op.weight" in sd:
        h = _run_sequential(sd, f"input_blocks.{i}", h, emb, ctx, b, d_head)
        if i == 0 and "init_attn.0.norm.weight" in sd:
            h = _run_sequential(sd, "init_attn", h, emb, ctx, b, d_head)
        hs.append(h)
        i += 1
    h = _run_sequential(sd, "middle_block", h, emb, ctx, b, d_head)
    j = 0
    while f"output_blocks.{j}.0.in_layers.0.weight" in sd:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_sequential(sd, f"output_blocks.{j}", h, emb, ctx, b, d_head)
        j += 1
    y = F.silu(_gn(sd, "out.0", h, 1e-5))
    y = F.conv2d(y, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
