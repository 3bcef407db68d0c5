"""First-stage Encoder on the HIP operators (the step right before the loop).

Mirrors reference lvdm/modules/networks/ae_modules.py -- ResnetBlock (153-213), AttnBlock
(21-88), Downsample (92-112), Encoder (366-475) -- with the same parameter names
(`first_stage_model.encoder.*` loads strictly).  It produces the latent moments and the five
hidden states that become the decoder's `ref_context`.

Same kernels as the decoder: implicit-GEMM 3x3 convolutions, fused GroupNorm+swish, the
single-head d=C attention as GEMM -> row softmax -> GEMM.  The stride-2 downsample with the
reference's asymmetric (0,1,0,1) zero pad is the `pad=0` gather mode of tc_gemm_bf16.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import ops
from .autoencoder_dualref import MemoryEfficientAttnBlock, Normalize
from .common import Act, PackedModule, ceil_to, f32, pack_conv3x3, pack_linear
from .openaimodel3d import _conv_geom


class ResnetBlock(PackedModule):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if temb_channels > 0 or conv_shortcut:
            raise NotImplementedError("ResnetBlock variant unused by the autoencoder config")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)

    def _pack(self):
        pk = {"g1": f32(self.norm1.weight), "b1": f32(self.norm1.bias),
              "w1": pack_conv3x3(self.conv1.weight), "cb1": f32(self.conv1.bias),
              "g2": f32(self.norm2.weight), "b2": f32(self.norm2.bias),
              "w2": pack_conv3x3(self.conv2.weight), "cb2": f32(self.conv2.bias)}
        if self.in_channels != self.out_channels:
            pk["ws"], pk["bs"] = pack_linear(self.nin_shortcut.weight), f32(self.nin_shortcut.bias)
        return pk

    def forward(self, act: Act) -> Act:
        pk = self.pk
        g_in, _, _ = _conv_geom(act, act.c)
        h = ops.groupnorm(act.rows, pk["g1"], pk["b1"], samples=act.frames, rows=act.hw, eps=1e-6, silu=True)
        h = ops.gemm(h, pk["w1"], pk["cb1"], conv=g_in)
        h = ops.groupnorm(h, pk["g2"], pk["b2"], samples=act.frames, rows=act.hw, eps=1e-6, silu=True)
        skip = act.rows if "ws" not in pk else ops.gemm(act.rows, pk["ws"], pk["bs"])
        g_out, _, _ = _conv_geom(act, self.out_channels)
        return act.like(ops.gemm(h, pk["w2"], pk["cb2"], conv=g_out, residual=skip))


class Downsample(PackedModule):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("average-pool Downsample is unused by the config")
        self.conv = nn.Conv2d(in_channels, in_channels, 3, stride=2, padding=0)

    def _pack(self):
        return {"w": pack_conv3x3(self.conv.weight), "b": f32(self.conv.bias)}

    def forward(self, act: Act) -> Act:
        ho, wo = (act.h + 1 - 3) // 2 + 1, (act.w + 1 - 3) // 2 + 1       # pad (0,1,0,1), 3x3, stride 2
        geom = dict(kind="3x3", frames=act.frames, cin=act.c, h_in=act.h, w_in=act.w, h_out=ho, w_out=wo,
                    stride=2, upsample=False, pad=0)
        return act.like(ops.gemm(act.rows, self.pk["w"], self.pk["b"], conv=geom), ho, wo)


class AttnBlock(MemoryEfficientAttnBlock):
    """Same arithmetic and parameter names as the decoder's single-head block."""


class _Level(nn.Module):
    pass


class Encoder(PackedModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if attn_resolutions or use_linear_attn:
            raise NotImplementedError("Encoder variant unused by the config")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            down = _Level()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def _pack(self):
        return {"wi": pack_conv3x3(self.conv_in.weight), "bi": f32(self.conv_in.bias),
                "og": f32(self.norm_out.weight), "ob": f32(self.norm_out.bias),
                "wo": pack_conv3x3(self.conv_out.weight), "bo": f32(self.conv_out.bias)}

    def encode_rows(self, x: torch.Tensor, out_w=None, out_b=None):
        """x: (N, 3, H, W) fp32 -> (moments rows fp32 [N*h*w, 2z], hidden Acts (4 levels + conv_in), h, w).
        `out_w/out_b`: packed replacement for conv_out (the autoencoder passes conv_out fused with its
        1x1 quant_conv)."""
        with ops.fp8_scope("decoder"):         # the first stage (pixels <-> latents) stays bf16 under TC_FP8
            return self._encode_rows(x, out_w, out_b)

    def _encode_rows(self, x, out_w, out_b):
        n, c, hh, ww = x.shape
        pk = self.pk
        cpad = ceil_to(c, 64)
        act = Act(ops.nchw_to_rows(x.float().reshape(n, c, 1, hh, ww), c_pad=cpad), n, 1, hh, ww)
        geom, _, _ = _conv_geom(act, cpad)
        act = act.like(ops.gemm(act.rows, pk["wi"], pk["bi"], conv=geom))
        first = act
        hidden: List[Act] = []
        for lvl in range(self.num_resolutions):
            for blk in self.down[lvl].block:
                act = blk(act)
            hidden.append(act)
            if lvl != self.num_resolutions - 1:
                act = self.down[lvl].downsample(act)
        hidden.append(first)
        act = self.mid.block_1(act)
        act = self.mid.attn_1(act)
        act = self.mid.block_2(act)
        hrows = ops.groupnorm(act.rows, pk["og"], pk["ob"], samples=act.frames, rows=act.hw, eps=1e-6, silu=True)
        geom, _, _ = _conv_geom(act, act.c)
        moments = ops.gemm(hrows, pk["wo"] if out_w is None else out_w, pk["bo"] if out_b is None else out_b,
                           conv=geom, out_f32=True)
        return moments, hidden, act.h, act.w

    def forward(self, x, return_hidden_states=False):
        """Reference call shape: (N, 3, H, W) -> (N, 2z, H/8, W/8) [, list of (N, C, H_l, W_l)]."""
        moments, hidden, h, w = self.encode_rows(x)
        n = x.shape[0]
        out = ops.rows_to_nchw(moments, c=moments.shape[1], b=n, t=1, h=h, w=w)[:, :, 0]
        if not return_hidden_states:
            return out
        hs = [ops.rows_to_nchw(a.rows, c=a.c, b=n, t=1, h=a.h, w=a.w)[:, :, 0] for a in hidden]
        return out, hs
