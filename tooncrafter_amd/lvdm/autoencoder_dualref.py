"""Dual-reference frame-aware VideoDecoder on the HIP operators.

Mirrors reference lvdm/models/autoencoder_dualref.py -- ResnetBlock (35-92),
MemoryEfficientAttnBlock (145-206), MemoryEfficientCrossAttentionWrapperFusion
(256-341), Combiner (343-368), Decoder (371-527), 3-D ResBlock (554-698),
VideoResBlock (846-911), AE3DConv (914-935), VideoDecoder (1121-1176) -- with the
same parameter names, so `first_stage_model.decoder.*` checkpoints load strictly.

Result-preserving restructurings:
  * the reference keys/values of the fusion attention are projected from the TWO
    reference frames once per clip and shared by all T frames (kv_bdiv = T); the
    reference repeats them T times (autoencoder_dualref.py:283-292);
  * `alpha*x3d + (1-alpha)*x2d` with x3d = x2d + conv(...) is evaluated as
    x2d + alpha*conv(...) in the epilogue of the last temporal conv;
  * the single-head d=512 mid attention is GEMM -> row softmax -> GEMM; V's bias
    is added after P*V (softmax rows sum to one);
  * nearest-x2 upsampling is folded into the following conv's gather.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn

from .. import ops
from .common import Act, PackedModule, SourceKey, ceil_to, f32, pack_conv3x3, pack_convt3, pack_linear
from .openaimodel3d import Upsample, _conv_geom


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class TimeStack(nn.Module):
    """Parameter container for the 3-D ResBlock (skip_t_emb): in_layers = [GN, SiLU, Conv3d],
    out_layers = [GN, SiLU, Dropout, Conv3d]."""

    def __init__(self, channels, dropout):
        super().__init__()
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0)))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0)))
        self.skip_connection = nn.Identity()


class VideoResBlock(PackedModule):
    def __init__(self, *, in_channels, out_channels=None, dropout=0.0, temb_channels=0, video_kernel_size=3,
                 alpha=0.0, merge_strategy="learned", conv_shortcut=False):
        super().__init__()
        if temb_channels > 0 or conv_shortcut or merge_strategy != "learned":
            raise NotImplementedError("VideoResBlock variant unused by the config")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)
        self.time_stack = TimeStack(out_channels, dropout)
        self.mix_factor = nn.Parameter(torch.Tensor([alpha]))

    def _pack(self):
        ts = self.time_stack
        pk = {"g1": f32(self.norm1.weight), "b1": f32(self.norm1.bias),
              "w1": pack_conv3x3(self.conv1.weight), "cb1": f32(self.conv1.bias),
              "g2": f32(self.norm2.weight), "b2": f32(self.norm2.bias),
              "w2": pack_conv3x3(self.conv2.weight), "cb2": f32(self.conv2.bias),
              "tg1": f32(ts.in_layers[0].weight), "tb1": f32(ts.in_layers[0].bias),
              "tw1": pack_convt3(ts.in_layers[2].weight), "tcb1": f32(ts.in_layers[2].bias),
              "tg2": f32(ts.out_layers[0].weight), "tb2": f32(ts.out_layers[0].bias),
              "tw2": pack_convt3(ts.out_layers[3].weight), "tcb2": f32(ts.out_layers[3].bias),
              "alpha": float(torch.sigmoid(self.mix_factor.detach().float()).item())}
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
        x2d = ops.gemm(h, pk["w2"], pk["cb2"], conv=g_out, residual=skip)
        # temporal ResBlock: fp32-statistics GroupNorm over (T, H, W) jointly, eps 1e-5
        gt = dict(kind="t3", frames=act.frames, t_len=act.t, cin=self.out_channels, h_out=act.h, w_out=act.w)
        h = ops.groupnorm(x2d, pk["tg1"], pk["tb1"], samples=act.b, rows=act.t * act.hw, eps=1e-5, silu=True)
        h = ops.gemm(h, pk["tw1"], pk["tcb1"], conv=gt)
        h = ops.groupnorm(h, pk["tg2"], pk["tb2"], samples=act.b, rows=act.t * act.hw, eps=1e-5, silu=True)
        out = ops.gemm(h, pk["tw2"], pk["tcb2"], conv=gt, out_scale=pk["alpha"], residual=x2d)
        return act.like(out)


class MemoryEfficientAttnBlock(PackedModule):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def _pack(self):
        return {"g": f32(self.norm.weight), "b": f32(self.norm.bias),
                "wq": pack_linear(self.q.weight), "bq": f32(self.q.bias),
                "wk": pack_linear(self.k.weight), "bk": f32(self.k.bias),
                "wv": pack_linear(self.v.weight), "bv": f32(self.v.bias),
                "wo": pack_linear(self.proj_out.weight), "bo": f32(self.proj_out.bias)}

    def forward(self, act: Act) -> Act:
        pk = self.pk
        c, l, f = act.c, act.hw, act.frames
        hn = ops.groupnorm(act.rows, pk["g"], pk["b"], samples=f, rows=l, eps=1e-6)
        q = ops.gemm(hn, pk["wq"], pk["bq"])
        k = ops.gemm(hn, pk["wk"], pk["bk"])
        # V^T per frame = Wv hn_f^T  (the GEMM with operand roles swapped): [F, C, L]
        vt = torch.empty((f * c, l), dtype=torch.bfloat16, device=hn.device)
        ops.gemm(pk["wv"], hn[:l], out=vt[:c], batch=f, stride_a=0, stride_w=l * c, stride_c=c * l)
        s = torch.empty((f * l, l), dtype=torch.float32, device=hn.device)
        ops.gemm(q[:l], k[:l], alpha=float(c) ** -0.5, out=s[:l], out_f32=True, batch=f,
                 stride_a=l * c, stride_w=l * c, stride_c=l * l)
        p = ops.softmax_rows(s)
        o = torch.empty((f * l, c), dtype=torch.bfloat16, device=hn.device)
        ops.gemm(p[:l], vt[:c], pk["bv"], out=o[:l], batch=f, stride_a=l * l, stride_w=c * l, stride_c=l * c)
        return act.like(ops.gemm(o, pk["wo"], pk["bo"], residual=act.rows))


class RefContext:
    """The five encoder hidden states of the first/last frame as bf16 rows `[B*2*H*W, C]`
    (row = (b*2 + l)*HW + p), plus the fusion blocks' projected K/V, reusable across decode calls.
    A new clip with the same geometry REFRESHES the buffers in place, so captured hipGraphs of the
    decoder keep pointing at valid data (same scheme as attention.ContextCache)."""

    def __init__(self, ref_context: List[torch.Tensor]):
        self.rows = []
        self.geom = []
        for r in ref_context:
            b, c, l, h, w = r.shape
            if l != 2:
                raise ValueError("ref_context tensors must be (B, C, 2, H, W)")
            self.rows.append(torch.empty((b * l * h * w, c), dtype=torch.bfloat16, device=r.device))
            self.geom.append((b, c, h, w))
        self.kv = {}           # id(module) -> (module, level, kv buffer)
        self.key = None
        self.refresh(ref_context)

    def matches(self, ref_context) -> bool:
        return len(ref_context) == len(self.geom) and all(
            tuple(r.shape) == (b, c, 2, h, w) and r.device == rows.device
            for r, (b, c, h, w), rows in zip(ref_context, self.geom, self.rows))

    def refresh(self, ref_context):
        for r, rows in zip(ref_context, self.rows):
            ops.nchw_to_rows(r.detach().float(), c_pad=r.shape[1], out=rows)
        for module, level, kv in self.kv.values():
            module.project_ref(self, level, kv)
        self.key = SourceKey(ref_context)      # strong references + versions: never data_ptr identity


class MemoryEfficientCrossAttentionWrapperFusion(PackedModule):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0, **kwargs):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("head dim 64 only")
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.norm = Normalize(query_dim)

    def _pack(self):
        return {"g": f32(self.norm.weight), "b": f32(self.norm.bias), "wq": pack_linear(self.to_q.weight),
                "wkv": pack_linear(torch.cat([self.to_k.weight, self.to_v.weight], 0)),
                "wo": pack_linear(self.to_out[0].weight), "bo": f32(self.to_out[0].bias)}

    def project_ref(self, ref: RefContext, level: int, out=None):
        return ops.gemm(ref.rows[level], self.pk["wkv"], out=out)               # [B*2*HW, 2*inner], once per clip

    def forward(self, act: Act, ref: RefContext, level: int) -> Act:
        pk = self.pk
        inner = self.heads * 64
        hit = ref.kv.get(id(self))
        if hit is None:
            hit = ref.kv[id(self)] = (self, level, self.project_ref(ref, level))
        kv = hit[2]
        hn = ops.groupnorm(act.rows, pk["g"], pk["b"], samples=act.frames, rows=act.hw, eps=1e-6)
        q = ops.gemm(hn, pk["wq"])
        a = ops.attention(q, kv[:, :inner], kv[:, inner:], batch=act.frames, heads=self.heads, lq=act.hw,
                          lk=2 * act.hw, kv_bdiv=act.t, scale=64 ** -0.5)
        return act.like(ops.gemm(a, pk["wo"], pk["bo"], residual=act.rows))


class Combiner(PackedModule):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 1, padding=0)

    def _pack(self):
        return {"w": pack_linear(self.conv.weight), "b": f32(self.conv.bias)}

    def forward(self, act: Act, ref: RefContext, level: int) -> Act:
        """x[:, first frame] += conv(ref first); x[:, last frame] += conv(ref last), in place on
        the rows of those two frames (the GEMM reads them as its residual and writes them back)."""
        pk = self.pk
        hw = act.hw
        ctx = ref.rows[level]
        for b in range(act.b):
            for l, tt in ((0, 0), (1, act.t - 1)):
                src = ctx[(b * 2 + l) * hw:(b * 2 + l + 1) * hw]
                dst = act.rows[(b * act.t + tt) * hw:(b * act.t + tt + 1) * hw]
                ops.gemm(src, pk["w"], pk["b"], residual=dst, out=dst)
        return act


class AE3DConv(nn.Conv2d):
    """Parameter container: Conv2d weight/bias + `time_mix_conv` Conv3d (3,1,1)."""

    def __init__(self, in_channels, out_channels, video_kernel_size=3, *args, **kwargs):
        super().__init__(in_channels, out_channels, *args, **kwargs)
        self.time_mix_conv = nn.Conv3d(out_channels, out_channels, kernel_size=(3, 1, 1), padding=(1, 0, 0))


class _Mid(nn.Module):
    pass


class VideoDecoder(PackedModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla-xformers", attn_level=[2, 3],
                 video_kernel_size=[3, 1, 1], alpha: float = 0.0, merge_strategy: str = "learned",
                 time_mode: str = "conv-only", **ignorekwargs):
        super().__init__()
        if time_mode != "conv-only" or attn_resolutions or give_pre_end or tanh_out or out_ch != 3:
            raise NotImplementedError("VideoDecoder variant unused by the config")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.attn_level = attn_level
        self.z_channels = z_channels
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = _Mid()
        mk = lambda cin, cout: VideoResBlock(in_channels=cin, out_channels=cout, dropout=dropout,
                                             video_kernel_size=video_kernel_size, alpha=alpha,
                                             merge_strategy=merge_strategy)
        self.mid.block_1 = mk(block_in, block_in)
        self.mid.attn_1 = MemoryEfficientAttnBlock(block_in)
        self.mid.block_2 = mk(block_in, block_in)
        self.up = nn.ModuleList()
        self.attn_refinement = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(mk(block_in, block_out))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
            if i_level in attn_level:
                self.attn_refinement.insert(0, MemoryEfficientCrossAttentionWrapperFusion(query_dim=block_in))
            else:
                self.attn_refinement.insert(0, Combiner(block_in))
        self.norm_out = Normalize(block_in)
        self.attn_refinement.append(Combiner(block_in))
        self.conv_out = AE3DConv(block_in, out_ch, video_kernel_size=video_kernel_size, kernel_size=3, stride=1, padding=1)
        self._ref_cache = None
        self._graphs = {}          # (z shape, scale, with refs, fp8 routing) -> static input/output + captured hipGraph
        self._graph_epoch = PackedModule.graph_epoch()
        self.use_hipgraph = os.environ.get("TC_HIPGRAPH", "1") != "0"

    def _pack(self):
        return {"wi": pack_conv3x3(self.conv_in.weight), "bi": f32(self.conv_in.bias),
                "og": f32(self.norm_out.weight), "ob": f32(self.norm_out.bias),
                "wo": pack_conv3x3(self.conv_out.weight), "bo": f32(self.conv_out.bias),
                "tw": f32(self.conv_out.time_mix_conv.weight).reshape(-1), "tb": f32(self.conv_out.time_mix_conv.bias)}

    def prepack(self):
        for m in self.modules():
            if isinstance(m, PackedModule):
                _ = m.pk
        return self

    def ref_cache(self, ref_context) -> RefContext:
        c = self._ref_cache
        if c is None or not c.matches(ref_context):
            c = self._ref_cache = RefContext(ref_context)
            self._graphs.clear()                       # captured graphs point at the old buffers
        elif c.key is None or not c.key.same(ref_context):
            c.refresh(ref_context)                     # new clip, same geometry: in place
        return c

    def reset_conditioning(self):
        """Clip boundary: the next decode re-reads the reference features whatever tensors carry them."""
        if self._ref_cache is not None:
            self._ref_cache.key = None

    def _apply(self, fn, recurse=True):               # .to(device): static buffers and graphs are stale
        self._ref_cache = None
        self._graphs = {}
        return super()._apply(fn, recurse)

    def decode_clip(self, z, ref_context, scale=1.0, probe=None):
        """z: (B, zc, T, h, w) fp32 latent -> (B, 3, T, 8h, 8w) fp32.  `scale` multiplies z on the
        way in (decode_core's 1/scale_factor).  `probe(name, act)` (parity tests only) sees the
        activation after the mid block and after each level's reference fusion.

        The ~1000 launches of one decode are captured in a hipGraph per (latent shape, scale) right after the
        first (eager) decode of that geometry and replayed afterwards (static input/output buffers; the
        reference rows and K/V are refreshed in place per clip): one host call per decode instead of ~1000."""
        with ops.fp8_scope("decoder"):         # TC_FP8 routing stops at the decoder unless TC_FP8_DECODER=1
            return self._decode_clip(z, ref_context, scale, probe)

    def _decode_clip(self, z, ref_context, scale, probe):
        ref = self.ref_cache(ref_context) if ref_context else None
        if not (self.use_hipgraph and z.is_cuda and probe is None and ops.backend().name == "hip"):
            return self._decode(z, ref, scale, probe)
        be = ops.backend()
        # the graph replays the kernels it recorded: the MXFP8 routing state is part of its identity, and so are the
        # packed weight tensors (a load_state_dict / invalidate() since capture frees them: drop every graph then)
        if self._graph_epoch != PackedModule.graph_epoch():
            self._graphs.clear()
            self._graph_epoch = PackedModule.graph_epoch()
        key = (tuple(z.shape), float(scale), ref is not None, getattr(be, "fp8", None), getattr(be, "fp8_decoder", None))
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"z": torch.empty(z.shape, dtype=torch.float32, device=z.device),
                                      "graph": None, "out": None, "calls": 0}
        st["z"].copy_(z)
        st["calls"] += 1
        if st["graph"] is None:
            # first decode of this geometry: run eagerly (weight packing, reference K/V projection), then record the
            # graph at once -- capture executes nothing, so the eager result is what this call returns
            y = self._decode(st["z"], ref, scale, None)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["out"] = self._decode(st["z"], ref, scale, None)
            st["graph"] = g
            return y
        st["graph"].replay()
        return st["out"].clone()

    def _decode(self, z, ref, scale, probe):
        b, zc, t, h, w = z.shape
        pk = self.pk
        cpad = ceil_to(zc, 64)
        act = Act(ops.nchw_to_rows(z.float(), c_pad=cpad, scale=scale), b, t, h, w)
        geom, _, _ = _conv_geom(act, cpad)
        act = act.like(ops.gemm(act.rows, pk["wi"], pk["bi"], conv=geom))
        act = self.mid.block_1(act)
        act = self.mid.attn_1(act)
        act = self.mid.block_2(act)
        if probe is not None:
            probe("mid", act)
        for lvl in reversed(range(self.num_resolutions)):
            for blk in self.up[lvl].block:
                act = blk(act)
            if ref is not None:
                act = self.attn_refinement[lvl](act, ref, lvl)
            if probe is not None:
                probe(f"level{lvl}", act)
            if lvl != 0:
                act = self.up[lvl].upsample(act)
        hrows = ops.groupnorm(act.rows, pk["og"], pk["ob"], samples=act.frames, rows=act.hw, eps=1e-6, silu=True)
        act = act.like(hrows)
        if ref is not None:
            act = self.attn_refinement[self.num_resolutions](act, ref, self.num_resolutions)
        geom, _, _ = _conv_geom(act, act.c)
        y = ops.gemm(act.rows, pk["wo"], pk["bo"], conv=geom, out_f32=True)        # [M, 3] fp32
        return ops.time_mix3(y, pk["tw"], pk["tb"], b=b, t=t, h=act.h, w_=act.w)

    def forward(self, z, ref_context=None, timesteps=None, **kwargs):
        """Reference call shape: z (B*T, zc, h, w) with kwargs timesteps=T -> (B*T, 3, H, W)."""
        t = timesteps if timesteps is not None else z.shape[0]
        bt, zc, h, w = z.shape
        z5 = z.reshape(bt // t, t, zc, h, w).permute(0, 2, 1, 3, 4)
        out = self.decode_clip(z5, ref_context)
        return out.permute(0, 2, 1, 3, 4).reshape(bt, 3, out.shape[-2], out.shape[-1])
