"""Spatio-temporal UNet on the HIP operators.

Mirrors reference lvdm/modules/networks/openaimodel3d.py: TimestepEmbedSequential
(30-48), Downsample (51-77), Upsample (80-106), ResBlock (109-236),
TemporalConvBlock (239-279), UNetModel (281-603) -- same constructor kwargs and
parameter names (including the reference's `temopral_conv` spelling), hence the
same state-dict keys.

Internal layout is channels-last bf16 rows `[B*T*H*W, C]` from the first conv to
the last; the only conversions are at the module boundary.  Every convolution is
an implicit GEMM (tc_gemm_bf16 gather modes), GroupNorm+SiLU is one fused
operator, and bias / timestep-embedding / residual adds ride in GEMM epilogues.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_SILU
from .attention import ContextCache, SpatialTransformer, TemporalTransformer
from .common import Act, CfgShare, PackedModule, ceil_to, f32, pack_conv3x3, pack_convt3, pack_linear


def conv_nd(dims, *args, **kwargs):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*args, **kwargs)


def normalization(channels, num_groups=32):
    return nn.GroupNorm(num_groups, channels)          # eps 1e-5; statistics in fp32 in the kernel


def _conv_geom(act: Act, cin: int, stride=1, upsample=False):
    h_out = act.h * 2 if upsample else (act.h + 2 - 3) // stride + 1
    w_out = act.w * 2 if upsample else (act.w + 2 - 3) // stride + 1
    return dict(kind="3x3", frames=act.frames, cin=cin, h_in=act.h, w_in=act.w, h_out=h_out, w_out=w_out,
                stride=stride, upsample=upsample), h_out, w_out


class TimestepBlock(nn.Module):
    pass


class Downsample(PackedModule):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2 or padding != 1:
            raise NotImplementedError("only the strided-conv Downsample of the config")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=1)

    def _pack(self):
        return {"w": pack_conv3x3(self.op.weight), "b": f32(self.op.bias)}

    def forward(self, act: Act) -> Act:
        geom, ho, wo = _conv_geom(act, act.c, stride=2)
        return act.like(ops.gemm(act.rows, self.pk["w"], self.pk["b"], conv=geom), ho, wo)


class Upsample(PackedModule):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2 or padding != 1:
            raise NotImplementedError("only the nearest-x2 + conv Upsample of the config")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=1)

    def _pack(self):
        return {"w": pack_conv3x3(self.conv.weight), "b": f32(self.conv.bias)}

    def forward(self, act: Act) -> Act:
        # nearest x2 is folded into the conv's row gather: the upsampled tensor never exists
        geom, ho, wo = _conv_geom(act, act.c, upsample=True)
        return act.like(ops.gemm(act.rows, self.pk["w"], self.pk["b"], conv=geom), ho, wo)


class TemporalConvBlock(PackedModule):
    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError("tempspatial_aware is unused by the config")
        out_channels = in_channels if out_channels is None else out_channels
        k, p = (3, 1, 1), (1, 0, 0)
        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_channels), nn.SiLU(), nn.Conv3d(in_channels, out_channels, k, padding=p))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))

    def _pack(self):
        pk = {}
        for i in range(1, 5):
            seq = getattr(self, f"conv{i}")
            pk[f"g{i}"], pk[f"b{i}"] = f32(seq[0].weight), f32(seq[0].bias)
            pk[f"w{i}"], pk[f"cb{i}"] = pack_convt3(seq[-1].weight), f32(seq[-1].bias)
        return pk

    def forward(self, act: Act, part=None) -> Act:
        """`part`: GroupNorm partial sums of act.rows from the GEMM that produced them (ops.GnPart) or None.  Every
        convolution here feeds the next GroupNorm, so it is asked for the statistics of what it stores (ABI 9): the
        norm then reads its input once instead of twice."""
        pk = self.pk
        y = act.rows
        geom = dict(kind="t3", frames=act.frames, t_len=act.t, cin=act.c, h_out=act.h, w_out=act.w)
        # GroupNorm + SiLU + convolution as ONE host operator (ops.gn_conv): two launches; round 5 measured the fused alternative -- a
        # statistics pass and a convolution that normalises its own operand
        gn = dict(samples=act.b, rows=act.t * act.hw, eps=1e-5, silu=True, conv=geom)
        for i in range(1, 5):
            if i < 4:
                y, part = ops.gn_conv(y, pk[f"g{i}"], pk[f"b{i}"], pk[f"w{i}"], pk[f"cb{i}"], part=part, gn_stats=True, **gn)
            else:
                y = ops.gn_conv(y, pk[f"g{i}"], pk[f"b{i}"], pk[f"w{i}"], pk[f"cb{i}"], part=part, residual=act.rows, **gn)
        return act.like(y)


class ResBlock(TimestepBlock, PackedModule):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, use_conv=False, up=False, down=False, use_temporal_conv=False,
                 tempspatial_aware=False):
        PackedModule.__init__(self)
        if use_scale_shift_norm or up or down or dims != 2 or use_conv:
            raise NotImplementedError("ResBlock variant unused by the config")
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = out_channels or channels
        self.use_temporal_conv = use_temporal_conv
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1,
                                                   spatial_aware=tempspatial_aware)
        self.emb_slice = None      # (offset, width) into the UNet-wide batched embedding projection

    def _pack(self):
        pk = {"g1": f32(self.in_layers[0].weight), "b1": f32(self.in_layers[0].bias),
              "w1": pack_conv3x3(self.in_layers[2].weight), "cb1": f32(self.in_layers[2].bias),
              "g2": f32(self.out_layers[0].weight), "b2": f32(self.out_layers[0].bias),
              "w2": pack_conv3x3(self.out_layers[3].weight), "cb2": f32(self.out_layers[3].bias)}
        if not isinstance(self.skip_connection, nn.Identity):
            pk["ws"], pk["bs"] = pack_linear(self.skip_connection.weight), f32(self.skip_connection.bias)
        return pk

    def forward(self, act: Act, emb_all: torch.Tensor) -> Act:
        """emb_all: fp32 [B, sum_cout] = every ResBlock's emb_layers(emb), computed in one GEMM."""
        pk = self.pk
        off, width = self.emb_slice
        geom1, _, _ = _conv_geom(act, ceil_to(act.c, 64))
        gn = dict(samples=act.frames, rows=act.hw, eps=1e-5, silu=True)
        # both convolutions feed a GroupNorm (out_layers' / the temporal block's first): they emit its statistics (ABI 9)
        h, part = ops.gn_conv(act.rows, pk["g1"], pk["b1"], pk["w1"], pk["cb1"], conv=geom1, row_bias=emb_all[:, off:off + width],
                              row_div=act.t * act.hw, gn_stats=True, prefetch_extra=(pk["ws"],) if "ws" in pk else (), **gn)
        skip = act.rows if "ws" not in pk else ops.gemm(act.rows, pk["ws"], pk["bs"])
        geom2, _, _ = _conv_geom(act, self.out_channels)
        if not self.use_temporal_conv:
            return act.like(ops.gn_conv(h, pk["g2"], pk["b2"], pk["w2"], pk["cb2"], conv=geom2, part=part, residual=skip, **gn))
        h, part = ops.gn_conv(h, pk["g2"], pk["b2"], pk["w2"], pk["cb2"], conv=geom2, part=part, residual=skip, gn_stats=True, **gn)
        return self.temopral_conv(act.like(h), part)


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, act: Act, emb_all, ctx: Optional[ContextCache] = None, share=None) -> Act:
        """`share` (common.CfgShare): under batched guidance the layers in front of the first cross-attention see the
        single-copy batch; `emb_all` is then taken from it (single-copy rows before the split, n-fold after)."""
        for j, layer in enumerate(self):
            def run(layer=layer, act=act):
                if isinstance(layer, ResBlock):
                    return layer(act, emb_all if share is None else share.emb())
                if isinstance(layer, SpatialTransformer):
                    return layer(act, ctx, share if share is not None and not share.done else None)
                return layer(act)                      # TemporalTransformer, InputConv, Downsample, Upsample
            # in front of the split a layer's result is the same for every guided pass (CfgShare.cached)
            act = run() if share is None or share.done else share.cached(("layer", id(self), j), run)
        return act


class InputConv(PackedModule):
    """`input_blocks.0.0`: conv3x3 in_channels -> model_channels.  Held as an nn.Conv2d-shaped
    parameter pair named weight/bias so the key is `input_blocks.0.0.weight`."""

    def __init__(self, cin, cout):
        super().__init__()
        ref = nn.Conv2d(cin, cout, 3, padding=1)
        self.weight = ref.weight
        self.bias = ref.bias
        self.cin_pad = ceil_to(cin, 64)

    def _pack(self):
        return {"w": pack_conv3x3(self.weight, self.cin_pad), "b": f32(self.bias)}

    def forward(self, act: Act) -> Act:
        geom, _, _ = _conv_geom(act, self.cin_pad)
        return act.like(ops.gemm(act.rows, self.pk["w"], self.pk["b"], conv=geom))


class UNetModel(PackedModule):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                 transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                 tempspatial_aware=False, temporal_attention=True, use_relative_position=True,
                 use_causal_attention=False, temporal_length=None, use_fp16=False, addition_attention=False,
                 temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        if num_head_channels == -1:
            raise NotImplementedError("set num_head_channels (the config uses 64)")
        if resblock_updown or dims != 2 or not conv_resample:
            raise NotImplementedError("UNet variant unused by the config")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.channel_mult, self.temporal_attention = channel_mult, temporal_attention
        self.addition_attention, self.temporal_length = addition_attention, temporal_length
        self.image_cross_attention, self.default_fs, self.fs_condition = image_cross_attention, default_fs, fs_condition
        self.dtype = torch.float32
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if fs_condition:
            self.fps_embedding = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def st(ch, heads):
            return SpatialTransformer(ch, heads, num_head_channels, depth=transformer_depth, context_dim=context_dim,
                                      use_linear=use_linear, use_checkpoint=use_checkpoint, disable_self_attn=False,
                                      video_length=temporal_length, image_cross_attention=image_cross_attention,
                                      image_cross_attention_scale_learnable=image_cross_attention_scale_learnable)

        def tt(ch, heads, linear=use_linear, causal=use_causal_attention):
            return TemporalTransformer(ch, heads, num_head_channels, depth=transformer_depth, context_dim=context_dim,
                                       use_linear=linear, use_checkpoint=use_checkpoint, only_self_att=True,
                                       causal_attention=causal, relative_position=use_relative_position,
                                       temporal_length=temporal_length)

        def rb(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, tempspatial_aware=tempspatial_aware,
                            use_temporal_conv=temporal_conv)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(InputConv(in_channels, model_channels))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(tt(model_channels, 8, linear=False, causal=False))
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [rb(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch, ch // num_head_channels))
                    if temporal_attention:
                        layers.append(tt(ch, ch // num_head_channels))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        layers = [rb(ch, ch), st(ch, ch // num_head_channels)]
        if temporal_attention:
            layers.append(tt(ch, ch // num_head_channels))
        layers.append(rb(ch, ch))
        self.middle_block = TimestepEmbedSequential(*layers)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [rb(ch + ich, mult * model_channels)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(st(ch, ch // num_head_channels))
                    if temporal_attention:
                        layers.append(tt(ch, ch // num_head_channels))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        # batched timestep-embedding projection: one GEMM for every ResBlock's emb_layers
        self._resblocks: List[ResBlock] = [m for m in self.modules() if isinstance(m, ResBlock)]
        off = 0
        for m in self._resblocks:
            m.emb_slice = (off, m.out_channels)
            off += m.out_channels
        self._emb_total = off
        self._ctx_cache: Optional[ContextCache] = None

    # ------------------------------------------------------------------ packing
    def _pack(self):
        pk = {"te_w0": pack_linear(self.time_embed[0].weight), "te_b0": f32(self.time_embed[0].bias),
              "emb_w": pack_linear(torch.cat([m.emb_layers[1].weight for m in self._resblocks], 0)),
              "emb_b": f32(torch.cat([m.emb_layers[1].bias for m in self._resblocks], 0)),
              "og": f32(self.out[0].weight), "ob": f32(self.out[0].bias),
              "ow": pack_conv3x3(self.out[2].weight), "ocb": f32(self.out[2].bias)}
        if self.fs_condition:
            # emb = time_embed[2](h_t) + fps_embedding[2](h_f) = [h_t | h_f] @ [W_t | W_f]^T + (b_t + b_f):
            # the sum of the two branches is ONE GEMM over the concatenated K
            pk.update({"fe_w0": pack_linear(self.fps_embedding[0].weight), "fe_b0": f32(self.fps_embedding[0].bias),
                       "e2_w": pack_linear(torch.cat([self.time_embed[2].weight, self.fps_embedding[2].weight], 1)),
                       "e2_b": f32(self.time_embed[2].bias + self.fps_embedding[2].bias)})
        else:
            pk.update({"e2_w": pack_linear(self.time_embed[2].weight), "e2_b": f32(self.time_embed[2].bias)})
        return pk

    def prepack(self):
        for m in self.modules():
            if isinstance(m, PackedModule):
                _ = m.pk
        return self

    # ------------------------------------------------------------------ conditioning
    def context_cache(self, context: torch.Tensor, t: int) -> ContextCache:
        """Conditioning rows + projected K/V.  Rebuilt when the shape changes, refreshed IN PLACE
        when only the values do (same buffers, so captured hipGraphs stay valid)."""
        c = self._ctx_cache
        if c is None or not c.matches(context, t):
            c = self._ctx_cache = ContextCache(context, t)
        elif not c.is_current(context):
            c.refresh(context)
        return c

    def reset_conditioning(self):
        """Clip boundary: the next forward re-reads the conditioning whatever tensor carries it."""
        if self._ctx_cache is not None:
            self._ctx_cache.invalidate()

    def _embedding(self, timesteps, fs, b):
        """silu(time_embed(t) + fps_embedding(fs)) pushed through every ResBlock's emb Linear:
        fp32 [B, sum_cout]."""
        pk = self.pk
        mc = self.model_channels
        ted = 4 * mc
        dev = timesteps.device
        def dense(v):          # a drop-in caller may hand over `t.expand(b)` or a strided view: the kernel reads [n] densely
            return (v if v.dtype == torch.int64 else v.to(torch.float32)).contiguous()
        te = ops.timestep_embedding(dense(timesteps), mc, ceil_to(mc, 8))
        hcat = torch.empty((b, 2 * ted if self.fs_condition else ted), dtype=torch.bfloat16, device=dev)
        ops.gemm(te, pk["te_w0"], pk["te_b0"], act=ACT_SILU, out=hcat[:, :ted])
        if self.fs_condition:
            if fs is None:
                fs = torch.full((b,), self.default_fs, dtype=torch.long, device=dev)
            fe = ops.timestep_embedding(dense(fs), mc, ceil_to(mc, 8))
            ops.gemm(fe, pk["fe_w0"], pk["fe_b0"], act=ACT_SILU, out=hcat[:, ted:])
        # emb_layers = SiLU -> Linear for all ResBlocks at once (openaimodel3d.py:168-174, 219)
        semb = ops.gemm(hcat, pk["e2_w"], pk["e2_b"], act=ACT_SILU)
        return ops.gemm(semb, pk["emb_w"], pk["emb_b"], out_f32=True)

    # ------------------------------------------------------------------ forward
    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, x_parts=None, replicas=1,
                branches=False, **kwargs):
        """x: (B, in_channels, T, H, W) fp32 (or `x_parts` = [x, c_concat] to skip the torch.cat of
        the hybrid conditioning); timesteps: [B] long; context: (B, 77+16T, Cc); fs: [B] long.
        Extra kwargs are swallowed like the reference does (openaimodel3d.py:548).

        `replicas` = n > 1 (batched classifier-free guidance, ddpm3d.apply_model_multi): x / timesteps / fs describe ONE
        copy of batch b, `context` all n * b; the result has batch n * b as if the copy had been repeated n times, but the
        layers in front of the first cross-attention run once (common.CfgShare).  With `branches` the n passes go their
        own ways behind that point, each at batch b on its own HIP stream, and the result is the LIST of their outputs."""
        if features_adapter is not None:
            raise NotImplementedError("features_adapter is unused by the inference path")
        parts = x_parts if x_parts is not None else [x]
        b, _, t, hh, ww = parts[0].shape
        if replicas > 1 and (context is None or context.shape[0] != replicas * b):
            raise ValueError(f"replicas={replicas}: context must carry {replicas * b} samples")
        cin_pad = ceil_to(self.in_channels, 64)
        rows = ops.nchw_to_rows(parts[0], parts[1] if len(parts) > 1 else None, c_pad=cin_pad)
        act = Act(rows, b, t, hh, ww)
        ctx = self.context_cache(context, t)
        emb_all = self._embedding(timesteps, fs, b)
        if replicas > 1 and branches:
            return self._forward_branches(act, emb_all, ctx, replicas)
        share = CfgShare(replicas, emb_all) if replicas > 1 else None
        return self._body(act, emb_all, ctx, share)

    def _body(self, act: Act, emb_all, ctx, share: Optional[CfgShare]):
        """Input blocks -> middle -> output blocks -> out conv, from the rows of the latent to the (B, C, T, H, W) result."""
        t = act.t
        hs = []
        for i, module in enumerate(self.input_blocks):
            act = module(act, emb_all, ctx, share)
            if i == 0 and self.addition_attention:
                act = self.init_attn(act, emb_all, ctx, share)     # a TemporalTransformer: no context, no embedding
            hs.append(act)
        if share is not None:
            if not share.done:
                raise NotImplementedError("replicas > 1 needs a cross-attention in the input blocks to part the passes at")
            emb_all = share.embn
        act = self.middle_block(act, emb_all, ctx)
        for module in self.output_blocks:
            skip = hs.pop()
            if skip.b != act.b:                                    # the skip taken in front of the split: repeat it now
                skip = share.expand(skip)
            act = act.like(ops.concat_rows(act.rows, skip.rows))
            act = module(act, emb_all, ctx)
        pk = self.pk
        h = ops.groupnorm(act.rows, pk["og"], pk["ob"], samples=act.frames, rows=act.hw, eps=1e-5, silu=True)
        geom, _, _ = _conv_geom(act, act.c)
        y = ops.gemm(h, pk["ow"], pk["ocb"], conv=geom, out_f32=True)
        return ops.rows_to_nchw(y, c=self.out_channels, b=act.b, t=t, h=act.h, w=act.w)

    def _forward_branches(self, act: Act, emb_all, ctx: ContextCache, n: int):
        """The n guided passes as n batch-b walks of the network that share what precedes the first cross-attention:
        pass 0 runs on the current stream and records the shared tensors; passes 1.. replay them and run on side streams
        that wait for the split point of pass 0 -- under hipGraph capture this becomes a fork / join inside the graph.
        The shared tensors live until every pass has been enqueued and the current stream waits for the side streams
        before anything is released or consumed, so no cross-stream lifetime bookkeeping is needed."""
        if getattr(self, "_cross", None) is None:
            self._cross = [m for m in self.modules() if getattr(m, "is_self", True) is False]
        for m in self._cross:                                      # K/V of every cross-attention exist BEFORE the fork (a first
            m.context_kv(ctx)                                      # call would otherwise project them mid-pass 0, on a stream
                                                                   # the other passes do not wait for)
        share = CfgShare(n, emb_all, branches=True)
        cuda = act.rows.is_cuda
        main = torch.cuda.current_stream(act.rows.device) if cuda else None
        side = self._side_streams(n - 1, act.rows.device) if cuda else []
        outs = []
        for k in range(n):
            share.begin(k)
            if k == 0 or not cuda:
                outs.append(self._body(act, emb_all, ctx.branch(k, act.b), share))
            else:
                side[k - 1].wait_event(share.fork)
                with torch.cuda.stream(side[k - 1]):
                    outs.append(self._body(act, emb_all, ctx.branch(k, act.b), share))
        for s in side:
            main.wait_stream(s)
        return outs

    def _side_streams(self, n: int, device):
        st = getattr(self, "_streams", None)
        if st is None or len(st) < n or st[0].device != device:
            st = self._streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        return st[:n]
