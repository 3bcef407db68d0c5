"""PyTorch-ROCm custom-op binding of the hot path: `torch.ops.tooncrafter.*` (csrc/torch_ops.cpp, TORCH_LIBRARY).

The same `extern "C"` entry points as the ctypes binding (ops.HipOps), registered as torch operators with CUDA(HIP)
and Meta implementations: current-stream pickup and output allocation happen in C++, and the ops can be shape-checked
/ traced on the `meta` device without a GPU.  `TorchLibOps` is a drop-in for `HipOps`; select it with
`TC_BINDING=torch` (or `ops.set_backend(TorchLibOps())`).  Calls that use features the functional op schema does not
carry (an explicit `out=` buffer, batched GEMMs, accumulate) go through the inherited ctypes methods.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import ACT_NONE
from .ops import HipOps

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtooncrafter_torch.so")
_loaded = False


def load():
    """Register the `tooncrafter` operator namespace (idempotent).  Raises if the library is missing."""
    global _loaded
    if _loaded:
        return torch.ops.tooncrafter
    if not os.path.exists(LIB_PATH):
        raise _lib.TooncrafterHipError(f"{LIB_PATH} not found: build it with `python -m tooncrafter_amd.build`")
    torch.ops.load_library(LIB_PATH)
    got = int(torch.ops.tooncrafter.abi_version())
    if got != _lib.TC_ABI_VERSION:
        # fatal, not a reason to fall back to ctypes: the namespace is registered by now (TORCH_LIBRARY runs at dlopen), so
        # `torch.ops.tooncrafter.*` would resolve to the stale layer for anybody who calls it directly
        raise _lib.TooncrafterAbiError(f"{LIB_PATH} was built against ABI {got}, this tree is ABI {_lib.TC_ABI_VERSION}: "
                                       "rebuild with `python -m tooncrafter_amd.build`")
    _loaded = True
    return torch.ops.tooncrafter


def conv_list(conv):
    """ops.gemm's `conv` dict -> the 11-integer form of the op schema ([] = linear)."""
    if conv is None:
        return []
    return [1 if conv["kind"] == "3x3" else 2, conv["cin"], conv["frames"], conv.get("t_len", 1),
            conv.get("h_in", conv["h_out"]), conv.get("w_in", conv["w_out"]), conv["h_out"], conv["w_out"],
            conv.get("stride", 1), 1 if conv.get("upsample", False) else 0, conv.get("pad", 1)]


class TorchLibOps(HipOps):
    """HipOps with the arithmetic operators dispatched through `torch.ops.tooncrafter`."""

    name = "hip"          # the same kernels: callers that assert the HIP backend see it as such
    binding = "torch"

    def __init__(self):
        super().__init__()
        self.t = load()

    def gemm(self, a, w, bias=None, *, act=ACT_NONE, residual=None, row_bias=None, row_div=0, alpha=1.0, out_scale=1.0,
             out=None, out_f32=False, conv=None, batch=1, stride_a=0, stride_w=0, stride_c=0, m=None, a_norm_eps=None,
             gn_stats=False):
        # fp8 routing and the ABI 9 producer statistics (a second result; only when TC_GN_PART=1 can produce them) live in HipOps.gemm
        if out is not None or batch != 1 or m is not None or self.fp8 is not None or (gn_stats and self.gn_part):
            return super().gemm(a, w, bias, act=act, residual=residual, row_bias=row_bias, row_div=row_div, alpha=alpha,
                                out_scale=out_scale, out=out, out_f32=out_f32, conv=conv, batch=batch, stride_a=stride_a,
                                stride_w=stride_w, stride_c=stride_c, m=m, a_norm_eps=a_norm_eps, gn_stats=gn_stats)
        res = self.t.gemm(a, w, bias, residual, row_bias, int(row_div), int(act), float(alpha), float(out_scale),
                          bool(out_f32), conv_list(conv), -1.0 if a_norm_eps is None else float(a_norm_eps))
        return (res, None) if gn_stats else res                 # ResBlock / TemporalConvBlock always ask; no producer statistics here

    def quant_mxfp8(self, x, k=None):
        return self.t.quant_mxfp8(x, int(x.shape[1] if k is None else k))

    def attention(self, q, k, v, *, batch, heads, lq, lk, kv_bdiv=1, out=None, accumulate=False, scale=None,
                  k2=None, v2=None, lk2=0, kv2_bdiv=1):
        if out is not None or accumulate:
            return super().attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv, out=out,
                                     accumulate=accumulate, scale=scale, k2=k2, v2=v2, lk2=lk2, kv2_bdiv=kv2_bdiv)
        return self.t.attention(q, k, v, batch, heads, lq, lk, kv_bdiv, float(64 ** -0.5 if scale is None else scale),
                                k2, v2, int(lk2), int(kv2_bdiv))

    def attention_temporal(self, qkv, *, b, t, hw, heads, scale=None):
        return self.t.attention_temporal(qkv, b, t, hw, heads, float(64 ** -0.5 if scale is None else scale))

    def ff_geglu_fused(self, x, w1, b1, w2, b2, *, ln_eps=None):
        return self.t.ff_geglu_fused(x, w1, b1, w2, b2, -1.0 if ln_eps is None else float(ln_eps))

    def temporal_attn_fused(self, x, wqkv, bqkv, wo, bo, *, b, t, hw, heads, ln_eps=None, scale=None):
        return self.t.temporal_attn_fused(x, wqkv, bqkv, wo, bo, int(b), int(t), int(hw), int(heads),
                                          -1.0 if ln_eps is None else float(ln_eps), float(64 ** -0.5 if scale is None else scale))

    def temporal_qkv_attn(self, x, wqkv, bqkv=None, *, b, t, hw, heads, scale=None, out=None):
        if out is not None:                                  # an explicit result buffer: the ctypes method (the op schema is functional)
            return super().temporal_qkv_attn(x, wqkv, bqkv, b=b, t=t, hw=hw, heads=heads, scale=scale, out=out)
        return self.t.temporal_qkv_attn(x, wqkv, bqkv, int(b), int(t), int(hw), int(heads), float(64 ** -0.5 if scale is None else scale))

    def groupnorm(self, x, gamma, beta, *, samples, rows, eps, silu=False, part=None, prefetch=None, prefetch_linear=False):
        if part is not None:
            return super().groupnorm(x, gamma, beta, samples=samples, rows=rows, eps=eps, silu=silu, part=part, prefetch=prefetch,
                                     prefetch_linear=prefetch_linear)
        pfl = self.prefetch_list(x.shape[0], prefetch, linear=prefetch_linear)   # ABI 12: the consumer's weights ride on the launch
        if pfl:
            return self.t.groupnorm_pf(x, gamma, beta, samples, rows, float(eps), bool(silu), pfl)
        return self.t.groupnorm(x, gamma, beta, samples, rows, float(eps), bool(silu))

    def layernorm(self, x, gamma, beta, eps=1e-5, mx_for=None, prefetch=None):
        if mx_for is not None and self.fp8 is not None:
            return super().layernorm(x, gamma, beta, eps, mx_for=mx_for, prefetch=prefetch)
        pfl = self.prefetch_list(x.shape[0], prefetch, linear=True)
        if pfl:
            return self.t.layernorm_pf(x, gamma, beta, float(eps), pfl)
        return self.t.layernorm(x, gamma, beta, float(eps))

    def ddim_step(self, x, e_cond, e_uncond, noise, *, cfg_scale, guidance_rescale, sqrt_ac, sqrt_1m_ac, sqrt_a_prev,
                  dir_coef, sigma, x0_rescale, want_x0=True, e_uncond_img=None, cfg_img=None):
        xp, x0 = self.t.ddim_step(x, e_cond, e_uncond, noise, e_uncond_img, float(cfg_scale),
                                  float(cfg_scale if cfg_img is None else cfg_img), float(guidance_rescale), float(sqrt_ac),
                                  float(sqrt_1m_ac), float(sqrt_a_prev), float(dir_coef), float(sigma), float(x0_rescale))
        return xp, (x0 if want_x0 else None)
