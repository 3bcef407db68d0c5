"""Oracle: first-stage Encoder + quant_conv + posterior, functional fp32.

Restates (reference paths under lvdm/):
  modules/networks/ae_modules.py:55-80     AttnBlock.forward (1 head, d = C, bmm + softmax)
  modules/networks/ae_modules.py:104-112   Downsample: pad (0,1,0,1) then conv 3x3 stride 2 pad 0
  modules/networks/ae_modules.py:157-175   ResnetBlock.forward (temb is None)
  modules/networks/ae_modules.py:432-475   Encoder.forward (return_hidden_states)
  models/autoencoder.py:100-110            AutoencoderKL.encode (quant_conv, posterior)
  distributions.py:24-40                   DiagonalGaussianDistribution (logvar clamp, sample)
  scripts/evaluation/inference.py:164-178  hidden states of the first and last frame only
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(sd, p, x, pad, stride=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad, stride=stride)


def resnet_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h)), 1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, 0)
    return x + h


def attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    n, c, hh, ww = x.shape
    h = _gn(sd, p + ".norm", x)
    q, k, v = (_conv(sd, f"{p}.{nm}", h, 0).reshape(n, c, hh * ww) for nm in "qkv")
    w_ = torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** -0.5)       # [n, hw_q, hw_k]
    w_ = torch.softmax(w_, dim=2)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(n, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", o, 0)


def encoder_forward(sd: SD, x: torch.Tensor, num_levels: int = 4, num_res_blocks: int = 2):
    """x: (N, 3, H, W) -> (moments-before-quant (N, 2z, H/8, W/8), [hidden states])."""
    hs = [_conv(sd, "conv_in", x, 1)]
    hidden = []
    for lvl in range(num_levels):
        for ib in range(num_res_blocks):
            hs.append(resnet_block(sd, f"down.{lvl}.block.{ib}", hs[-1]))
        hidden.append(hs[-1])
        if lvl != num_levels - 1:
            hs.append(_conv(sd, f"down.{lvl}.downsample.conv", F.pad(hs[-1], (0, 1, 0, 1)), 0, stride=2))
    hidden.append(hs[0])
    h = resnet_block(sd, "mid.block_1", hs[-1])
    h = attn_block(sd, "mid.attn_1", h)
    h = resnet_block(sd, "mid.block_2", h)
    h = _conv(sd, "conv_out", _swish(_gn(sd, "norm_out", h)), 1)
    return h, hidden


def encode(sd_fs: SD, x: torch.Tensor, noise: torch.Tensor | None = None, scale_factor: float = 0.18215, **kw):
    """`first_stage_model.*` state dict.  Returns (z = scale_factor * sample, mean, logvar, hidden states)."""
    enc = {k[len("encoder."):]: v for k, v in sd_fs.items() if k.startswith("encoder.")}
    h, hidden = encoder_forward(enc, x, **kw)
    moments = F.conv2d(h, sd_fs["quant_conv.weight"], sd_fs["quant_conv.bias"])
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return scale_factor * z, mean, logvar, hidden


def first_last_hidden(hidden: List[torch.Tensor], t: int) -> List[torch.Tensor]:
    """(b t) c h w -> (b, c, 2, h, w): frames 0 and t-1 of every clip."""
    out = []
    for hid in hidden:
        n, c, hh, ww = hid.shape
        h5 = hid.reshape(n // t, t, c, hh, ww).permute(0, 2, 1, 3, 4)
        out.append(torch.cat([h5[:, :, 0:1], h5[:, :, -1:]], dim=2))
    return out
