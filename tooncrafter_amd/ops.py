"""Tensor-level operators of the hot path, bound to the HIP C ABI.

PyTorch is plumbing here: it owns device memory (caching allocator, so outputs
are stream-ordered and hipGraph-capturable) and the current stream.  Every
arithmetic operation is a `tc_*` entry point of libtooncrafter_hip.so.

Data convention: activations are bf16 "rows" tensors `[M, C]` (channels-last:
M = B*T*H*W), possibly column-sliced views of a wider buffer (row stride =
`stride(0)`, unit column stride).  Weights are packed once by
`tooncrafter_amd.lvdm.packing`.

The functions below dispatch to the active backend object.  The only backend
the package ships is `HipOps`; tests may install a CPU emulation of the same
contract (tests/emu_ops.py) to exercise the host logic without a GPU.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU, GATHER_CONV3x3, GATHER_CONVT3,
                   GATHER_LINEAR, TcAttnParams, TcDdimParams, TcFfParams, TcGemmMxParams, TcGemmParams, TcTbParams, TcTqaParams)

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows_view(t: torch.Tensor, dtype=BF16):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype:
        raise ValueError(f"expected a 2-D {dtype} rows tensor with unit column stride, got "
                         f"{tuple(t.shape)} {t.dtype} strides {t.stride()}")
    if not t.is_cuda:
        raise _lib.TooncrafterHipError("HIP operator called with a CPU tensor: the product path is GPU-only")
    return t


def _dev(t: Optional[torch.Tensor], dtype, what: str, contiguous: bool = True):
    """Pointer of a device operand, or raise: a CPU pointer handed to a kernel faults the GPU (and kills
    the process) instead of raising, so every tensor whose data_ptr() crosses the C ABI is checked here."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.TooncrafterHipError(f"{what}: expected a CUDA tensor, got "
                                       f"{getattr(t, 'device', type(t))}: the product path is GPU-only")
    if t.dtype != dtype:
        raise ValueError(f"{what}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{what}: must be contiguous")
    return t.data_ptr()


class MxRows:
    """An activation that exists only as MXFP8 (e4m3 bytes + E8M0 block scales): what `layernorm(..., mx_for=...)`
    hands to the one GEMM that consumes it when the fp8 routing takes that GEMM."""

    def __init__(self, q, scales, k):
        self.q, self.scales, self.k = q, scales, k
        self.shape = (q.shape[0], k)
        self.device = q.device


class GnPart:
    """Partial GroupNorm sums a GEMM emitted beside its output (ABI 9): `sums` fp32 [row blocks, 2, N] (sum, sum of
    squares of the rounded outputs per block of `rows` rows and per column), `of` the output tensor they describe."""
    __slots__ = ("sums", "rows", "of")

    def __init__(self, sums, rows, of):
        self.sums, self.rows, self.of = sums, rows, of


class HipOps:
    """The product backend: ctypes calls into libtooncrafter_hip.so."""

    name = "hip"

    def __init__(self):
        import os
        self.lib = _lib.load()
        self._ws = {}
        # BASELINE.json configs[4]: TC_FP8 routes eligible GEMMs (see _fp8_eligible) through MXFP8 operands -- weights
        # quantised once, activations by tc_quant_mxfp8 in front of each GEMM.  TC_FP8=1 = "linear": the wide-N
        # projections (qkv, GEGLU), which carry most of the time gain at 2x the bf16 error of a UNet forward;
        # "all" adds the 3x3 / temporal convolutions (10x the bf16 error, profiles/r02_fp8_error_by_layer_class.txt);
        # "conv" | "conv3" | "convt" select those alone.
        mode = os.environ.get("TC_FP8", "0").lower()
        self.fp8 = None if mode in ("0", "", "off") else ("linear" if mode in ("1", "on") else mode)
        self.fp8_min_k = int(os.environ.get("TC_FP8_MIN_K", "640"))
        self.fp8_min_n = int(os.environ.get("TC_FP8_MIN_N", "1280"))
        self.fp8_max_cin = int(os.environ.get("TC_FP8_MAX_CIN", "1280"))
        self.fp8_min_cin = int(os.environ.get("TC_FP8_MIN_CIN", "0"))
        self.fp8_n_over_k = float(os.environ.get("TC_FP8_N_OVER_K", "2"))
        self.fp8_min_m = 1024
        self.fp8_decoder = os.environ.get("TC_FP8_DECODER", "0") == "1"
        self.fp8_fuse_ln = os.environ.get("TC_FP8_FUSE_LN", "1") != "0"     # LayerNorm emits MXFP8 for its fp8 consumer
        self.fp8_calls = {"mx": 0, "bf16": 0}
        self._wq = {}
        # LayerNorm -> consumer GEMM fusion (ABI 8, gemm_ln_eligible); TC_FUSE_LN=0 keeps the separate launch (A/B runs)
        self.fuse_ln = os.environ.get("TC_FUSE_LN", "1") != "0"
        # ABI 9: GroupNorm statistics from the producing GEMM.  OFF by default: measured end to end (profiles/
        # r04_gn_part_ab.txt, same lease, alternating runs) GroupNorm loses 0.3-0.4 ms per guided forward (4.71 -> 4.39,
        # 6.08 -> 5.72 on a slow lease) and the producing convolutions gain 0.4-1.1 ms (their epilogues fold 160 rows per
        # column and end on two block barriers): 7.79 vs 7.79 frames/s on one lease, 6.63 vs 6.75 on the other
        # (a call that carries gn_part keeps the implicit-GEMM kernels: the halo-patch route, the default for the 3x3
        # convolutions of levels 0-2 since round 5, emits no statistics -- so TC_GN_PART=1 also moves those convolutions back,
        # which is the routing the figures above were measured on)
        self.gn_part = os.environ.get("TC_GN_PART", "0") == "1"
        # ABI 12: the norm in front of a GEMM streams that GEMM's weights into the Infinity Cache (tc_groupnorm_pf /
        # tc_layernorm_pf).  Inside a forward W is always cold (2.9 GB of parameters per forward, 256 MB of cache); that
        # costs the 1280-channel levels 7-29 % of their GEMM time and nothing where W is small beside A
        # (profiles/r05_cold_operand_probe.txt, r05_prefetch_premise_probe.txt) -- hence the row bound.  TC_PREFETCH=0: never.
        self.prefetch_on = os.environ.get("TC_PREFETCH", "1") != "0"
        self.prefetch_max_rows = int(os.environ.get("TC_PREFETCH_MAX_ROWS", "8192"))
        self.prefetch_min_bytes = 1 << 20
        self.prefetch_max_bytes = 96 << 20

    # ------------------------------------------------------------------ workspace
    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        # stream-ordered scratch from the caching allocator (capture-safe)
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)

    # ------------------------------------------------------------------ GEMM family
    def gemm(self, a, w, bias=None, *, act=ACT_NONE, residual=None, row_bias=None, row_div=0,
             alpha=1.0, out_scale=1.0, out=None, out_f32=False, conv=None, batch=1,
             stride_a=0, stride_w=0, stride_c=0, m=None, a_norm_eps=None, gn_stats=False):
        """out[M, N'] = act(alpha * gather(a) @ w^T + bias + row_bias[m // row_div]) * out_scale + residual.

        gn_stats (ABI 9): the result is (out, GnPart | None) -- when the kernel this problem runs on can emit them
        (tc_gemm_gn_rows), per-row-block column sums of the rounded outputs, which `groupnorm(..., part=)` takes instead
        of reading the tensor once more for its statistics; None otherwise (the caller passes it on all the same).

        a_norm_eps (ABI 8): LayerNorm of the rows of `a` as a prologue of the product -- (x - mean) * rsqrt(var + eps)
        over the K columns; gamma / beta must already be folded into w / bias (lvdm.common.fold_layernorm).  Only for
        problems `gemm_ln_eligible` accepts (the weight-stationary K = 320 kernel).

        a: rows tensor (for conv modes the SOURCE rows, lda = a.stride(0));
        w: [N, K] bf16 contiguous; conv: None | dict(kind='3x3'|'t3', frames, t_len, h_in, w_in,
        h_out, w_out, stride, upsample, cin); batch>1: a/w/out are the first batch item views
        and stride_* the element strides between items."""
        mx_a = a if isinstance(a, MxRows) else None
        if mx_a is not None:
            if conv is not None or out is not None or batch != 1 or self.fp8 is None:
                raise ValueError("gemm: an MXFP8 activation feeds one plain linear GEMM of the fp8 route")
            a = mx_a.q                                   # shape / device bookkeeping below; never read as bf16
        else:
            a = _rows_view(a)
        if w.dtype != BF16 or w.dim() != 2 or w.stride(1) != 1:
            raise ValueError("w must be a [N, K] bf16 tensor with unit column stride")
        _dev(w, BF16, "gemm: w", contiguous=False)
        n, k = w.shape
        n_out = n // 2 if act == ACT_GEGLU else n
        p = TcGemmParams()
        if conv is None:
            mm = a.shape[0] if m is None else m
            p.gather = GATHER_LINEAR
            if a.shape[1] < k:
                raise ValueError(f"A has {a.shape[1]} columns, weight K = {k}")
        else:
            kind = conv["kind"]
            p.gather = GATHER_CONV3x3 if kind == "3x3" else GATHER_CONVT3
            p.cin = conv["cin"]
            p.frames = conv["frames"]
            p.t_len = conv.get("t_len", 1)
            p.h_out, p.w_out = conv["h_out"], conv["w_out"]
            p.h_in, p.w_in = conv.get("h_in", conv["h_out"]), conv.get("w_in", conv["w_out"])
            p.stride = conv.get("stride", 1)
            p.upsample = 1 if conv.get("upsample", False) else 0
            p.pad = conv.get("pad", 1)
            mm = p.frames * p.h_out * p.w_out
            need_rows = p.frames * p.h_in * p.w_in
            if a.shape[0] < need_rows:
                raise ValueError(f"conv source has {a.shape[0]} rows, geometry needs {need_rows}")
            if a.shape[1] < p.cin or k != (9 if kind == "3x3" else 3) * p.cin:
                raise ValueError(f"conv source has {a.shape[1]} channels / weight K = {k}, geometry says cin = {p.cin}")
        if out is None:
            out = torch.empty((mm, n_out), dtype=torch.float32 if out_f32 else BF16, device=a.device)
        else:
            _rows_view(out, torch.float32 if out_f32 else BF16)
            if out.shape[0] < mm or out.shape[1] != n_out:
                raise ValueError(f"out shape {tuple(out.shape)} != ({mm}, {n_out})")
        p.a, p.w, p.c = a.data_ptr(), w.data_ptr(), out.data_ptr()
        if bias is not None:
            if bias.numel() != n:
                raise ValueError("bias must be fp32 [N]")
            p.bias = _dev(bias, torch.float32, "gemm: bias")
        if row_bias is not None:
            if row_bias.dtype != torch.float32 or row_bias.dim() != 2 or row_bias.stride(1) != 1 \
                    or row_bias.shape[1] != n:
                raise ValueError("row_bias must be fp32 [R, N] with unit column stride")
            if row_div <= 0 or row_bias.shape[0] * row_div < mm:
                raise ValueError("row_bias / row_div do not cover M")
            p.row_bias = _dev(row_bias, torch.float32, "gemm: row_bias", contiguous=False)
            p.ldrb = row_bias.stride(0)
        if residual is not None:
            _rows_view(residual)
            if residual.shape[1] != n_out or residual.shape[0] < mm:
                raise ValueError("residual shape mismatch")
            p.residual = residual.data_ptr()
            p.ldr = residual.stride(0)
        p.m, p.n, p.k = mm, n, k
        p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), out.stride(0)
        p.row_div = row_div
        p.alpha, p.out_scale = float(alpha), float(out_scale)
        p.act, p.out_f32 = act, 1 if out_f32 else 0
        p.batch = batch
        p.stride_a, p.stride_w, p.stride_c = stride_a, stride_w, stride_c
        if a_norm_eps is not None:
            p.a_norm, p.a_norm_eps = 1, float(a_norm_eps)
            if mx_a is not None or self.fp8 is not None and self._fp8_eligible(p, conv is not None, n_out, batch):
                raise ValueError("gemm: a_norm_eps and the MXFP8 route exclude each other")
            if not self.lib.tc_gemm_ws_eligible(C.byref(p)):
                raise ValueError("gemm: a_norm_eps needs a problem the weight-stationary kernel takes (gemm_ln_eligible)")
        if self.fp8 is not None:
            if self._fp8_eligible(p, conv is not None, n_out, batch):
                self.fp8_calls["mx"] += 1
                self._gemm_mx(p, a, w, mx_a)
                return (out, None) if gn_stats else out
            if mx_a is not None:
                raise ValueError("gemm: MXFP8 activation handed to a GEMM the fp8 route does not take")
            self.fp8_calls["bf16"] += 1
        nbytes = self.lib.tc_gemm_workspace(C.byref(p))          # > 0 only for split-K candidates (low-res layers)
        if nbytes > 0:
            ws = self._workspace(nbytes, a.device)
            p.workspace, p.workspace_bytes = ws.data_ptr(), nbytes
        part = None
        if gn_stats and self.gn_part and out.is_contiguous():
            prow = self.lib.tc_gemm_gn_rows(C.byref(p))         # 160 / 128 / 0: the routing's own answer
            if prow > 0:
                part = GnPart(torch.empty(((mm + prow - 1) // prow, 2, n), dtype=torch.float32, device=a.device), prow, out)
                p.gn_part = part.sums.data_ptr()
        _lib.check(self.lib.tc_gemm_bf16(C.byref(p), _stream()), "tc_gemm_bf16")
        return (out, part) if gn_stats else out

    def gemm_ln_eligible(self, m, n, k, *, geglu=False, lda=None) -> bool:
        """Would `gemm(a[m, k], w[n, k], act=GEGLU if geglu, a_norm_eps=...)` be accepted, i.e. may the caller drop the
        LayerNorm launch in front of this projection?  ONE rule, the library's own (tc_gemm_ws_eligible); the MXFP8
        route keeps its LayerNorm -> MXFP8 fusion instead."""
        if not self.fuse_ln:
            return False
        p = TcGemmParams()
        p.gather, p.m, p.n, p.k = GATHER_LINEAR, int(m), int(n), int(k)
        p.lda, p.ldw = int(lda if lda is not None else k), int(k)
        p.act = ACT_GEGLU if geglu else ACT_NONE
        p.ldc = p.ldr = n // 2 if geglu else n
        p.alpha, p.out_scale, p.batch = 1.0, 1.0, 1
        if not self.lib.tc_gemm_ws_eligible(C.byref(p)):
            return False
        return not (self.fp8 is not None and self._fp8_eligible(p, False, p.ldc, 1))

    # ------------------------------------------------------------------ fused level-0 feed-forward (ABI 9)
    def _ff_params(self, m, c, hidden, ldx, ldo, ln_eps):
        p = TcFfParams()
        p.m, p.c, p.hidden, p.ldx, p.ldo = int(m), int(c), int(hidden), int(ldx), int(ldo)
        p.ln, p.ln_eps = (0, 0.0) if ln_eps is None else (1, float(ln_eps))
        return p

    def ff_fused_eligible(self, m, c, hidden, *, ldx=None) -> bool:
        """Would `ff_geglu_fused(x[m, c], w1[2 hidden, c], ...)` be accepted?  The library's own rule
        (tc_ff_geglu_fused_eligible: c = 320, hidden = 1280, TC_FF_FUSED != 0); never on the MXFP8 route, whose GEGLU
        projection is a different kernel with its own numerics."""
        if self.fp8 is not None:
            return False
        p = self._ff_params(m, c, hidden, c if ldx is None else ldx, c, 1e-5)
        return bool(self.lib.tc_ff_geglu_fused_eligible(C.byref(p)))

    def ff_geglu_fused(self, x, w1, b1, w2, b2, *, ln_eps=None):
        """out = x + w2 . GEGLU(w1 . LN(x) + b1) + b2 as ONE launch (reference attention.py:415-442 behind norm3): the hidden
        tensor never reaches HBM.  w1 / b1: the GEGLU projection exactly as `gemm(..., act=ACT_GEGLU)` takes it (with the
        LayerNorm's affine half folded in when ln_eps is given); ln_eps=None: x is taken as already normalised."""
        m, c = x.shape
        hidden = w2.shape[1]
        _dev(x, BF16, "ff_geglu_fused x", contiguous=False)
        for t, dt, what in ((w1, BF16, "w1"), (w2, BF16, "w2"), (b1, torch.float32, "b1"), (b2, torch.float32, "b2")):
            _dev(t, dt, "ff_geglu_fused " + what)
        if x.stride(1) != 1 or tuple(w1.shape) != (2 * hidden, c) or tuple(w2.shape) != (c, hidden) or b1.numel() != 2 * hidden \
                or b2.numel() != c:
            raise ValueError("ff_geglu_fused: x [m, c] rows, w1 [2 hidden, c], b1 [2 hidden], w2 [c, hidden], b2 [c]")
        out = torch.empty((m, c), dtype=BF16, device=x.device)
        p = self._ff_params(m, c, hidden, x.stride(0), c, ln_eps)
        p.x, p.w1, p.b1, p.w2, p.b2, p.out = (x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                              out.data_ptr())
        _lib.check(self.lib.tc_ff_geglu_fused(C.byref(p), _stream()), "tc_ff_geglu_fused")
        return out

    # ------------------------------------------------------------------ fused level-0 temporal self-attention (ABI 9)
    def _tb_params(self, b, t, hw, c, heads, ldx, ldo, ln_eps, scale):
        p = TcTbParams()
        p.b, p.t, p.hw, p.c, p.heads, p.ldx, p.ldo = int(b), int(t), int(hw), int(c), int(heads), int(ldx), int(ldo)
        p.ln, p.ln_eps = (0, 0.0) if ln_eps is None else (1, float(ln_eps))
        p.scale = float(64 ** -0.5 if scale is None else scale)
        return p

    def temporal_attn_fused_eligible(self, *, b, t, hw, c, heads, ldx=None) -> bool:
        """Would `temporal_attn_fused` be accepted?  The library's own rule (tc_temporal_attn_fused_eligible: c = 320, 5 heads,
        16 frames, hw % 8 == 0, and TC_TB_FUSED=1 -- since round 6 the default route of level 0 is LayerNorm + `temporal_qkv_attn` +
        output projection, which measured ahead); never on the MXFP8 route."""
        if self.fp8 is not None:
            return False
        p = self._tb_params(b, t, hw, c, heads, c if ldx is None else ldx, c, 1e-5, None)
        return bool(self.lib.tc_temporal_attn_fused_eligible(C.byref(p)))

    def temporal_attn_fused(self, x, wqkv, bqkv, wo, bo, *, b, t, hw, heads, ln_eps=None, scale=None):
        """out = x + wo . Attn_frames(wqkv . LN(x) + bqkv) + bo as ONE launch (reference attention.py:81-144 over the frames of
        a pixel, TemporalTransformer attention.py:365-412): the qkv tensor and the attention output never reach HBM.  wqkv:
        the fused projection `gemm` takes in front of `attention_temporal` (LayerNorm's affine half folded in when ln_eps is
        given, bqkv = its bias; zeros otherwise)."""
        m, c = x.shape
        _dev(x, BF16, "temporal_attn_fused x", contiguous=False)
        for tns, dt, what in ((wqkv, BF16, "wqkv"), (wo, BF16, "wo"), (bqkv, torch.float32, "bqkv"), (bo, torch.float32, "bo")):
            _dev(tns, dt, "temporal_attn_fused " + what)
        if x.stride(1) != 1 or m != b * t * hw or tuple(wqkv.shape) != (3 * c, c) or tuple(wo.shape) != (c, c) \
                or bqkv.numel() != 3 * c or bo.numel() != c or c != heads * 64:
            raise ValueError("temporal_attn_fused: x [b*t*hw, c] rows, wqkv [3c, c], bqkv [3c], wo [c, c], bo [c], c = heads * 64")
        out = torch.empty((m, c), dtype=BF16, device=x.device)
        p = self._tb_params(b, t, hw, c, heads, x.stride(0), c, ln_eps, scale)
        p.x, p.wqkv, p.bqkv, p.wo, p.bo, p.out = (x.data_ptr(), wqkv.data_ptr(), bqkv.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                                  out.data_ptr())
        _lib.check(self.lib.tc_temporal_attn_fused(C.byref(p), _stream()), "tc_temporal_attn_fused")
        return out

    # ------------------------------------------------------------------ temporal qkv projection + attention, one launch (ABI 13)
    def _tqa_params(self, b, t, hw, c, heads, ldx, ldo, scale):
        p = TcTqaParams()
        p.b, p.t, p.hw, p.c, p.heads, p.ldx, p.ldo = int(b), int(t), int(hw), int(c), int(heads), int(ldx), int(ldo)
        p.scale = float(64 ** -0.5 if scale is None else scale)
        return p

    def temporal_qkv_attn_eligible(self, *, b, t, hw, c, heads, ldx=None) -> bool:
        """Would `temporal_qkv_attn` be accepted?  The library's own rule (tc_temporal_qkv_attn_eligible: 16 frames,
        c = heads * 64, hw % 8 == 0, TC_QKV_ATTN != 0); never on the MXFP8 route (which quantises the projection's input)."""
        if self.fp8 is not None:
            return False
        p = self._tqa_params(b, t, hw, c, heads, c if ldx is None else ldx, c, None)
        return bool(self.lib.tc_temporal_qkv_attn_eligible(C.byref(p)))

    def temporal_qkv_attn(self, x, wqkv, bqkv=None, *, b, t, hw, heads, scale=None, out=None):
        """Attn_frames(x . wqkv^T + bqkv) -> [rows, c] as ONE launch: `gemm(x, wqkv, bqkv)` + `attention_temporal` without the
        [rows, 3c] tensor between them (reference attention.py:96-134 over the frames of a pixel, TemporalTransformer
        attention.py:365-412).  x: the projection's input (LayerNorm output) rows; the result feeds to_out."""
        m, c = x.shape
        _dev(x, BF16, "temporal_qkv_attn x", contiguous=False)
        _dev(wqkv, BF16, "temporal_qkv_attn wqkv")
        if bqkv is not None:
            _dev(bqkv, torch.float32, "temporal_qkv_attn bqkv")
        if x.stride(1) != 1 or m != b * t * hw or tuple(wqkv.shape) != (3 * c, c) or c != heads * 64 \
                or (bqkv is not None and bqkv.numel() != 3 * c):
            raise ValueError("temporal_qkv_attn: x [b*t*hw, c] rows, wqkv [3c, c], bqkv [3c] or None, c = heads * 64")
        if out is None:
            out = torch.empty((m, c), dtype=BF16, device=x.device)
        elif tuple(out.shape) != (m, c) or out.stride(1) != 1:
            raise ValueError("temporal_qkv_attn: out must be [b*t*hw, c] rows")
        else:
            _dev(out, BF16, "temporal_qkv_attn out", contiguous=False)
        p = self._tqa_params(b, t, hw, c, heads, x.stride(0), out.stride(0), scale)
        p.x, p.wqkv, p.bqkv, p.out = x.data_ptr(), wqkv.data_ptr(), (None if bqkv is None else bqkv.data_ptr()), out.data_ptr()
        _lib.check(self.lib.tc_temporal_qkv_attn(C.byref(p), _stream()), "tc_temporal_qkv_attn")
        return out

    # ------------------------------------------------------------------ MXFP8 GEMM path (configs[4])
    def quant_mxfp8(self, x, k=None):
        """bf16 rows [R, >=k] -> (e4m3 bytes [R, k], E8M0 scales [R, ceil(k/128)*4]); one scale per 32 K elements."""
        x = _rows_view(x)
        k = x.shape[1] if k is None else k
        if k % 32 or x.shape[1] < k:
            raise ValueError(f"quant_mxfp8: k = {k} must be a multiple of 32 and <= {x.shape[1]} columns")
        rows, lds = x.shape[0], (k + 127) // 128 * 4
        q = torch.empty((rows, k), dtype=torch.uint8, device=x.device)
        sc = torch.empty((rows, lds), dtype=torch.uint8, device=x.device)
        _lib.check(self.lib.tc_quant_mxfp8(x.data_ptr(), rows, k, x.stride(0), q.data_ptr(), q.stride(0),
                                           sc.data_ptr(), lds, _stream()), "tc_quant_mxfp8")
        return q, sc

    @contextlib.contextmanager
    def fp8_scope(self, name):
        """Modules whose output precision matters more than their time opt out of the fp8 routing: the VAE decoder
        writes the pixels (measured at full size: 1.3e-1 from the fp32 oracle with MXFP8 convolutions against 1.3e-2
        in bf16, for 3 ms per decode) -- TC_FP8_DECODER=1 opts it back in."""
        old = self.fp8
        if name == "decoder" and not self.fp8_decoder:
            self.fp8 = None
        try:
            yield
        finally:
            self.fp8 = old

    def _weight_mx(self, w):
        """Quantised copy of a packed weight matrix, made once per weight tensor (held while the tensor lives)."""
        import weakref
        ent = self._wq.get(id(w))
        if ent is None or ent[0]() is not w or ent[1] != w._version:
            q, sc = self.quant_mxfp8(w if w.stride(0) % 8 == 0 else w.contiguous())
            ent = (weakref.ref(w), w._version, q, sc)
            if id(w) not in self._wq:
                weakref.finalize(w, self._wq.pop, id(w), None)
            self._wq[id(w)] = ent
        return ent[2], ent[3]

    def _fp8_eligible(self, p, is_conv, n_out, batch) -> bool:
        if batch != 1 or n_out % 8 or p.k % 32 or p.m < self.fp8_min_m:
            return False
        # measured (profiles/r02_mx_gemm_bench.txt): with the activation quantiser in front, fp8 pays on the 3x3 /
        # temporal convolutions up to cin = 1280 (one quantisation feeds 9 / 3 taps) and on linear layers whose N is
        # large against K (qkv, GEGLU: N >= 2 K -- exactly the consumers of a LayerNorm, which then emits MXFP8
        # itself); short-K or narrow-N linear layers (out-projections, ff2: slower in fp8 even before the quantiser
        # pass, profiles/r02_mx_gemm_bench.txt) stay bf16
        if is_conv:
            kind = "conv3" if p.gather == GATHER_CONV3x3 else "convt"
            return self.fp8 in ("all", "conv", kind) and p.cin % 64 == 0 and self.fp8_min_cin <= p.cin <= self.fp8_max_cin
        return self.fp8 in ("all", "linear") and p.k >= self.fp8_min_k and p.n >= self.fp8_min_n \
            and p.n >= self.fp8_n_over_k * p.k

    def _gemm_mx(self, p, a, w, mx_a=None):
        kc = p.cin if p.gather != GATHER_LINEAR else p.k
        rows = p.frames * p.h_in * p.w_in if p.gather == GATHER_CONV3x3 else p.m
        if mx_a is not None:
            if mx_a.k != kc:
                raise ValueError(f"gemm: MXFP8 activation has K = {mx_a.k}, weight K = {kc}")
            aq, asc = mx_a.q, mx_a.scales
        else:
            aq, asc = self.quant_mxfp8(a[:rows], kc)
        wq, wsc = self._weight_mx(w)
        px = TcGemmMxParams()
        px.g = p
        px.g.a, px.g.lda = aq.data_ptr(), aq.stride(0)
        px.g.w, px.g.ldw = wq.data_ptr(), wq.stride(0)
        px.a_scale, px.lda_s = asc.data_ptr(), asc.stride(0)
        px.w_scale, px.ldw_s = wsc.data_ptr(), wsc.stride(0)
        _lib.check(self.lib.tc_gemm_mxfp8(C.byref(px), _stream()), "tc_gemm_mxfp8")

    # ------------------------------------------------------------------ attention
    def attention(self, q, k, v, *, batch, heads, lq, lk, kv_bdiv=1, out=None, accumulate=False, scale=None,
                  k2=None, v2=None, lk2=0, kv2_bdiv=1):
        """q: [batch*lq, heads*64] rows view; k, v: [(batch//kv_bdiv)*lk, heads*64] rows views.
        out[batch*lq, heads*64] (+)= softmax(q k^T * scale) v per (batch, head)
        [+ softmax(q k2^T * scale) v2 when a second key/value set is given: two softmaxes, one launch]."""
        q, k, v = _rows_view(q), _rows_view(k), _rows_view(v)
        hd = heads * 64
        if q.shape != (batch * lq, hd) or k.shape[1] != hd or v.shape[1] != hd:
            raise ValueError("attention: head layout mismatch")
        kvb = (batch + kv_bdiv - 1) // kv_bdiv
        if k.shape[0] != kvb * lk or v.shape[0] != kvb * lk:
            raise ValueError("attention: K/V rows != kv_batches * lk")
        if out is None:
            if accumulate:
                raise ValueError("accumulate needs out")
            out = torch.empty((batch * lq, hd), dtype=BF16, device=q.device)
        else:
            _rows_view(out)
        p = TcAttnParams()
        p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
        p.batch, p.heads, p.lq, p.lk = batch, heads, lq, lk
        p.q_ss, p.k_ss, p.v_ss, p.o_ss = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        p.q_sb, p.k_sb, p.v_sb, p.o_sb = lq * q.stride(0), lk * k.stride(0), lk * v.stride(0), lq * out.stride(0)
        p.kv_bdiv = kv_bdiv
        p.accumulate = 1 if accumulate else 0
        p.scale = float(scale if scale is not None else 64 ** -0.5)
        if k2 is not None:
            k2, v2 = _rows_view(k2), _rows_view(v2)
            kvb2 = (batch + kv2_bdiv - 1) // kv2_bdiv
            if lk2 <= 0 or k2.shape != (kvb2 * lk2, hd) or v2.shape != (kvb2 * lk2, hd):
                raise ValueError("attention: second K/V set must be [(batch//kv2_bdiv)*lk2, heads*64]")
            p.k2, p.v2, p.lk2, p.kv2_bdiv = k2.data_ptr(), v2.data_ptr(), lk2, kv2_bdiv
            p.k2_ss, p.v2_ss = k2.stride(0), v2.stride(0)
            p.k2_sb, p.v2_sb = lk2 * k2.stride(0), lk2 * v2.stride(0)
        _lib.check(self.lib.tc_attn_d64(C.byref(p), _stream()), "tc_attn_d64")
        return out

    def attention_temporal(self, qkv, *, b, t, hw, heads, scale=None):
        """qkv: contiguous [b*t*hw, 3*heads*64] with row = (b*T + t)*HW + p -> [b*t*hw, heads*64]."""
        qkv = _rows_view(qkv)
        cdim = heads * 64
        if not qkv.is_contiguous() or qkv.shape != (b * t * hw, 3 * cdim):
            raise ValueError("attention_temporal: qkv must be contiguous [b*t*hw, 3*C]")
        out = torch.empty((b * t * hw, cdim), dtype=BF16, device=qkv.device)
        _lib.check(self.lib.tc_attn_temporal(qkv.data_ptr(), out.data_ptr(), b, t, hw, heads,
                                             float(scale if scale is not None else 64 ** -0.5), _stream()),
                   "tc_attn_temporal")
        return out

    # ------------------------------------------------------------------ norms
    def prefetch_list(self, x_rows: int, tensors, linear: bool = False):
        """Which of `tensors` (the packed weights of the GEMMs that consume a norm's output) the norm launch should
        stream ahead of them: ONE rule for both bindings.  Only when the consumer is a small-M GEMM (its weights are then a
        large share of its traffic), only tensors worth a request, at most TC_PREFETCH_MAX of them / 96 MB.
        `linear`: the tensors are [N, K] weights of plain linear layers over these x_rows rows.  With the MXFP8 routing on
        (opt-in, configs[4]) a consumer it takes reads a cached QUANTISED copy of W (`_weight_mx`), not this bf16 tensor:
        streaming the bf16 one would be up to 96 MB of traffic and cache pollution for data nobody reads, so
        those are left out -- per tensor for linear consumers, wholesale when the routing also takes convolutions."""
        if not self.prefetch_on or not tensors or x_rows > self.prefetch_max_rows:
            return []
        if self.fp8 is not None and not linear and self.fp8 != "linear":
            return []
        out, total = [], 0
        for t in tensors:
            if t is None or not isinstance(t, torch.Tensor) or not t.is_contiguous() or t.data_ptr() % 16:
                continue                                              # (a hint: what the ABI would refuse is simply not streamed)
            if self.fp8 is not None and linear and t.dim() == 2:
                probe = TcGemmParams()
                probe.m, probe.n, probe.k, probe.gather = x_rows, t.shape[0], t.shape[1], GATHER_LINEAR
                if self._fp8_eligible(probe, False, t.shape[0], 1) or self._fp8_eligible(probe, False, t.shape[0] // 2, 1):
                    continue                                          # (a GEGLU weight's N_out is N / 2: either answer means "taken")
            nbytes = t.numel() * t.element_size()
            if nbytes < self.prefetch_min_bytes or total + nbytes > self.prefetch_max_bytes:
                continue
            out.append(t)
            total += nbytes
            if len(out) == _lib.TC_PREFETCH_MAX:
                break
        return out

    @staticmethod
    def _prefetch_struct(tensors):
        pf = _lib.TcPrefetch()
        for i, t in enumerate(tensors):
            if not t.is_cuda:
                raise _lib.TooncrafterHipError("prefetch: a CPU tensor in a norm's prefetch list: the product path is GPU-only")
            pf.ptr[i], pf.bytes[i] = t.data_ptr(), t.numel() * t.element_size()
        pf.n = len(tensors)
        return pf

    def groupnorm(self, x, gamma, beta, *, samples, rows, eps, silu=False, part=None, prefetch=None, prefetch_linear=False):
        """x: contiguous [samples*rows, C]; statistics over (rows, C/32) per (sample, group).  `part` (a GnPart from
        the gemm that produced x, or None): statistics from the producer's partial sums -- one pass over x, not two.
        `prefetch` (ABI 12): packed weights of the GEMMs that read the result (see prefetch_list)."""
        x = _rows_view(x)
        c = x.shape[1]
        if not x.is_contiguous() or x.shape[0] != samples * rows:
            raise ValueError("groupnorm: x must be contiguous [samples*rows, C]")
        if gamma.numel() != c or beta.numel() != c:
            raise ValueError("groupnorm: gamma/beta must have C elements")
        gp, bp = _dev(gamma, torch.float32, "groupnorm: gamma"), _dev(beta, torch.float32, "groupnorm: beta")
        y = torch.empty_like(x)
        nbytes = self.lib.tc_groupnorm_workspace(samples, rows, c)
        ws = self._workspace(nbytes, x.device)
        if part is not None and part.of is x and rows % part.rows == 0 and part.sums.shape[2] == c \
                and part.sums.shape[0] * part.rows == samples * rows:
            _lib.check(self.lib.tc_groupnorm_part(x.data_ptr(), y.data_ptr(), gp, bp, part.sums.data_ptr(), part.rows, samples,
                                                  rows, c, float(eps), 1 if silu else 0, ws.data_ptr(), nbytes, _stream()),
                       "tc_groupnorm_part")
            return y
        pfl = self.prefetch_list(x.shape[0], prefetch, linear=prefetch_linear)
        if pfl:
            pf = self._prefetch_struct(pfl)
            _lib.check(self.lib.tc_groupnorm_pf(x.data_ptr(), y.data_ptr(), gp, bp, samples, rows, c, float(eps), 1 if silu else 0,
                                                ws.data_ptr(), nbytes, C.byref(pf), _stream()), "tc_groupnorm_pf")
            return y
        _lib.check(self.lib.tc_groupnorm(x.data_ptr(), y.data_ptr(), gp, bp, samples,
                                         rows, c, float(eps), 1 if silu else 0, ws.data_ptr(), nbytes, _stream()),
                   "tc_groupnorm")
        return y

    def gn_conv(self, x, gamma, beta, w, bias=None, *, samples, rows, eps, conv, silu=True, part=None, prefetch_extra=(), **kw):
        """conv(act(GroupNorm(x))) with the epilogue of `gemm` (**kw: row_bias / row_div / residual / act / gn_stats / out_f32):
        the reference's pair lvdm/basics.py:76-87 -> nn.Conv2d / nn.Conv3d (openaimodel3d.py:154,179,255-266) as ONE host
        operator, two launches.  (Round 5 measured the alternative -- the convolution normalising its own operand, ABI 10 --
        at -4 % per clip and removed it: DESIGN.md 5.6.)  `part`: see groupnorm."""
        # ABI 12: the norm's launch brings the convolution's weights (and `prefetch_extra`: those of a GEMM right behind it,
        # e.g. a ResBlock's 1x1 skip convolution) into the Infinity Cache where that pays (prefetch_list)
        h = self.groupnorm(x, gamma, beta, samples=samples, rows=rows, eps=eps, silu=silu, part=part,
                           prefetch=[w, *prefetch_extra])
        return self.gemm(h, w, bias, conv=conv, **kw)

    def layernorm(self, x, gamma, beta, eps=1e-5, mx_for=None, prefetch=None):
        """`mx_for` = (N, N_out) of the packed weight of the ONE linear GEMM that consumes the result: when the fp8
        route takes that GEMM the row leaves as MXFP8 (an `MxRows`; the quantiser fused into its producer).
        `prefetch` (ABI 12): packed weights of the GEMMs behind the norm (see prefetch_list)."""
        x = _rows_view(x)
        if not x.is_contiguous():
            raise ValueError("layernorm: x must be contiguous")
        if gamma.numel() != x.shape[1] or beta.numel() != x.shape[1]:
            raise ValueError("layernorm: gamma/beta must have C elements")
        gp, bp = _dev(gamma, torch.float32, "layernorm: gamma"), _dev(beta, torch.float32, "layernorm: beta")
        if mx_for is not None and self.fp8 is not None and self.fp8_fuse_ln:
            probe = TcGemmParams()
            probe.m, probe.n, probe.k, probe.gather = x.shape[0], mx_for[0], x.shape[1], GATHER_LINEAR
            if self._fp8_eligible(probe, False, mx_for[1], 1):
                rows, k = x.shape
                lds = (k + 127) // 128 * 4
                q = torch.empty((rows, k), dtype=torch.uint8, device=x.device)
                sc = torch.empty((rows, lds), dtype=torch.uint8, device=x.device)
                _lib.check(self.lib.tc_layernorm_mxfp8(x.data_ptr(), q.data_ptr(), k, sc.data_ptr(), lds, gp, bp, rows, k,
                                                       float(eps), _stream()), "tc_layernorm_mxfp8")
                return MxRows(q, sc, k)
        y = torch.empty_like(x)
        pfl = self.prefetch_list(x.shape[0], prefetch, linear=True)
        if pfl:
            pf = self._prefetch_struct(pfl)
            _lib.check(self.lib.tc_layernorm_pf(x.data_ptr(), y.data_ptr(), gp, bp, x.shape[0], x.shape[1], float(eps),
                                                C.byref(pf), _stream()), "tc_layernorm_pf")
            return y
        _lib.check(self.lib.tc_layernorm(x.data_ptr(), y.data_ptr(), gp, bp,
                                         x.shape[0], x.shape[1], float(eps), _stream()), "tc_layernorm")
        return y

    def softmax_rows(self, s, n=None, causal_period=0):
        """fp32 scores [rows, ld] -> bf16 probabilities [rows, ld]: softmax over the first `n` columns
        (default: all), the padding columns [n, ld) written as zeros; `causal_period` > 0: row r attends
        columns 0..(r mod period) only."""
        if s.dtype != torch.float32 or s.dim() != 2 or not s.is_contiguous() or not s.is_cuda:
            raise ValueError("softmax_rows: contiguous fp32 CUDA [rows, n]")
        ld = s.shape[1]
        n = ld if n is None else int(n)
        p = torch.empty(s.shape, dtype=BF16, device=s.device)
        _lib.check(self.lib.tc_softmax_rows(s.data_ptr(), p.data_ptr(), s.shape[0], n, ld, ld, ld,
                                            int(causal_period), _stream()), "tc_softmax_rows")
        return p

    # ------------------------------------------------------------------ layout / elementwise
    def nchw_to_rows(self, x0, x1=None, *, c_pad, scale=1.0, out=None):
        """(B, C0, T, H, W) [+ (B, C1, T, H, W)] fp32 -> bf16 rows [(b t h w), c_pad] (into `out` when given)."""
        if x0.dtype != torch.float32 or x0.dim() != 5 or not x0.is_cuda:
            raise ValueError("nchw_to_rows: fp32 CUDA (B, C, T, H, W)")
        x0 = x0.contiguous()
        b, c0, t, h, w = x0.shape
        c1 = 0
        if x1 is not None:
            x1 = x1.contiguous()
            if not x1.is_cuda or x1.dtype != torch.float32 or x1.shape[0] != b or tuple(x1.shape[2:]) != (t, h, w):
                raise ValueError("nchw_to_rows: second tensor shape mismatch")
            c1 = x1.shape[1]
        if out is None:
            out = torch.empty((b * t * h * w, c_pad), dtype=BF16, device=x0.device)
        elif not out.is_contiguous() or _rows_view(out).shape != (b * t * h * w, c_pad):
            raise ValueError("nchw_to_rows: out must be a contiguous [(b t h w), c_pad] bf16 tensor")
        _lib.check(self.lib.tc_nchw_to_rows(x0.data_ptr(), c0, _ptr(x1), c1, out.data_ptr(), c_pad, b, t, h * w,
                                            float(scale), _stream()), "tc_nchw_to_rows")
        return out

    def rows_to_nchw(self, rows, *, c, b, t, h, w):
        if rows.dim() != 2 or rows.stride(1) != 1 or rows.dtype not in (BF16, torch.float32) or not rows.is_cuda:
            raise ValueError("rows_to_nchw: 2-D bf16/fp32 CUDA rows")
        out = torch.empty((b, c, t, h, w), dtype=torch.float32, device=rows.device)
        _lib.check(self.lib.tc_rows_to_nchw(rows.data_ptr(), 1 if rows.dtype == torch.float32 else 0,
                                            rows.stride(0), out.data_ptr(), c, b, t, h * w, _stream()),
                   "tc_rows_to_nchw")
        return out

    def concat_rows(self, a, b):
        a, b = _rows_view(a), _rows_view(b)
        if not a.is_contiguous() or not b.is_contiguous() or a.shape[0] != b.shape[0]:
            raise ValueError("concat_rows: contiguous inputs with equal rows")
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=BF16, device=a.device)
        _lib.check(self.lib.tc_concat_rows(a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], out.data_ptr(),
                                           a.shape[0], _stream()), "tc_concat_rows")
        return out

    def timestep_embedding(self, t, dim, ld=None):
        """t: fp32 or int64 [n] -> bf16 [n, ld] = [cos | sin | 0 pad] (int64: converted to fp32 in the kernel, like the
        reference's `timesteps[:, None].float()`, utils_diffusion.py:19-23)."""
        if t.dtype not in (torch.float32, torch.int64) or t.dim() != 1 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("timestep_embedding: contiguous fp32 / int64 CUDA [n]")
        ld = dim if ld is None else ld
        out = torch.empty((t.shape[0], ld), dtype=BF16, device=t.device)
        fn = self.lib.tc_timestep_embedding if t.dtype == torch.float32 else self.lib.tc_timestep_embedding_i64
        _lib.check(fn(t.data_ptr(), out.data_ptr(), t.shape[0], dim, ld, _stream()), "tc_timestep_embedding")
        return out

    def repeat_rows(self, x, n):
        """[rows, C] -> [n * rows, C]: n copies one after the other (x.repeat(n, 1) as one launch of this library)."""
        if x.dim() != 2 or not x.is_cuda or not x.is_contiguous() or (x.shape[1] * x.element_size()) % 16:
            raise ValueError("repeat_rows: contiguous CUDA [rows, C] with 16-byte rows")
        out = torch.empty((n * x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.tc_repeat_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1] * x.element_size(), n, _stream()),
                   "tc_repeat_rows")
        return out

    def silu_to_bf16(self, x):
        if x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
            raise ValueError("silu_to_bf16: contiguous fp32 CUDA")
        y = torch.empty(x.shape, dtype=BF16, device=x.device)
        _lib.check(self.lib.tc_silu_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()),
                   "tc_silu_f32_to_bf16")
        return y

    def time_mix3(self, rows, w, bias, *, b, t, h, w_):
        """rows: fp32 [b*t*h*w_, ld>=3]; w: fp32 [3,3,3,1,1]; -> (b, 3, t, h, w_) fp32."""
        if rows.dtype != torch.float32 or rows.dim() != 2 or rows.stride(1) != 1 or not rows.is_cuda:
            raise ValueError("time_mix3: fp32 CUDA rows")
        if w.numel() != 27 or bias.numel() != 3:
            raise ValueError("time_mix3: w must have 27 elements, bias 3")
        wp, bp = _dev(w, torch.float32, "time_mix3: w"), _dev(bias, torch.float32, "time_mix3: bias")
        out = torch.empty((b, 3, t, h, w_), dtype=torch.float32, device=rows.device)
        _lib.check(self.lib.tc_time_mix3(rows.data_ptr(), rows.stride(0), wp, bp,
                                         out.data_ptr(), b, t, h * w_, _stream()), "tc_time_mix3")
        return out

    def video_to_uint8(self, video):
        """(b, 3, t, h, w) fp32 in [-1, 1] -> (b, t, h, w, 3) uint8: clamp, (x+1)/2*255, truncate, permute
        (scripts/evaluation/inference.py:148-153), on the device."""
        if video.dtype != torch.float32 or video.dim() != 5 or video.shape[1] != 3 or not video.is_cuda:
            raise ValueError("video_to_uint8: fp32 CUDA (b, 3, t, h, w)")
        video = video.contiguous()
        b, _, t, h, w = video.shape
        out = torch.empty((b, t, h, w, 3), dtype=torch.uint8, device=video.device)
        _lib.check(self.lib.tc_video_to_u8(video.data_ptr(), out.data_ptr(), b, t, h * w, _stream()), "tc_video_to_u8")
        return out

    def ddim_step(self, x, e_cond, e_uncond, noise, *, cfg_scale, guidance_rescale, sqrt_ac, sqrt_1m_ac,
                  sqrt_a_prev, dir_coef, sigma, x0_rescale, want_x0=True, e_uncond_img=None, cfg_img=None):
        """One fused DDIM update.  With `e_uncond_img` the guidance is the three-way form of
        ddim_multiplecond.py:236 (cfg_img defaults to cfg_scale like the reference)."""
        for tns in (x, e_cond, e_uncond, noise, e_uncond_img):
            if tns is not None and (tns.dtype != torch.float32 or not tns.is_contiguous() or not tns.is_cuda):
                raise ValueError("ddim_step: contiguous fp32 CUDA tensors")
        b = x.shape[0]
        n = x.numel() // b
        x_prev = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        p = TcDdimParams()
        p.x, p.e_cond, p.e_uncond, p.noise = x.data_ptr(), e_cond.data_ptr(), _ptr(e_uncond), _ptr(noise)
        p.x_prev, p.pred_x0 = x_prev.data_ptr(), _ptr(x0)
        p.b, p.n = b, n
        p.cfg_scale, p.guidance_rescale = float(cfg_scale), float(guidance_rescale)
        p.sqrt_ac, p.sqrt_1m_ac, p.sqrt_a_prev = float(sqrt_ac), float(sqrt_1m_ac), float(sqrt_a_prev)
        p.dir_coef, p.sigma, p.x0_rescale = float(dir_coef), float(sigma), float(x0_rescale)
        p.e_uncond_img = _ptr(e_uncond_img)
        p.cfg_img = float(cfg_scale if cfg_img is None else cfg_img)
        nbytes = self.lib.tc_ddim_workspace(b)
        ws = self._workspace(nbytes, x.device)
        _lib.check(self.lib.tc_ddim_step(C.byref(p), ws.data_ptr(), nbytes, _stream()), "tc_ddim_step")
        return x_prev, x0


_backend = None


_binding_fallback = None


def binding_fallback():
    """None, or why the default custom-op binding was NOT taken and ctypes stepped in (bench.py prints it: a line that says
    `binding: ctypes` by accident must be tellable from one that was asked for)."""
    return _binding_fallback


def backend():
    """The operator set behind `ops.gemm(...)` etc.  Two bindings over the ONE C ABI (include/tooncrafter_hip.h):
      TC_BINDING=torch (default since round 5): PyTorch-ROCm custom ops, `torch.ops.tooncrafter.*` (csrc/torch_ops.cpp,
        TORCH_LIBRARY with CUDA + Meta keys) -- the binding BASELINE.json's north_star names;
      TC_BINDING=ctypes: the same entry points through ctypes.
    Bit-identical (tests/test_gpu_torch_ops.py) and equally fast (bench.py `binding_ab`; both replay hipGraphs).  Either way
    libtooncrafter_hip.so is mandatory: no kernel library, no backend (HipOps raises).  If only the custom-op layer
    (libtooncrafter_torch.so, host C++) is missing, the ctypes binding takes over with one line on stderr -- unless TC_BINDING=torch
    was asked for explicitly, which then fails."""
    global _backend
    if _backend is None:
        import os
        import sys
        want = os.environ.get("TC_BINDING", "").lower()
        if want in ("", "torch"):
            _lib.load()                                              # the kernel library first: its absence is fatal either way
            from . import torch_ops
            try:
                torch_ops.load()
            except _lib.TooncrafterAbiError:                         # a STALE op library: fatal (its namespace is registered already)
                raise
            except Exception as e:                                   # noqa: BLE001  (missing / unloadable op library)
                if want == "torch":
                    raise
                global _binding_fallback
                _binding_fallback = f"{type(e).__name__}: {e}"
                sys.stderr.write(f"[tooncrafter_amd] custom-op layer unavailable ({_binding_fallback}); using the ctypes binding\n")
                _backend = HipOps()
            else:
                _backend = torch_ops.TorchLibOps()
        elif want == "ctypes":
            _backend = HipOps()   # raises if libtooncrafter_hip.so is missing: no fallback
        else:
            raise ValueError(f"TC_BINDING={want!r}: expected 'torch' or 'ctypes'")
    return _backend


@contextlib.contextmanager
def fp8_scope(name):
    """`with ops.fp8_scope("decoder"):` -- see HipOps.fp8_scope; a no-op for backends without an fp8 path."""
    fn = getattr(backend(), "fp8_scope", None)
    if fn is None:
        yield
    else:
        with fn(name):
            yield


def set_backend(b):
    """Test hook: install another implementation of the operator contract."""
    global _backend
    prev, _backend = _backend, b
    return prev


def __getattr__(name):   # ops.gemm(...) -> backend().gemm(...)
    if name.startswith("_"):
        raise AttributeError(name)
    return getattr(backend(), name)
