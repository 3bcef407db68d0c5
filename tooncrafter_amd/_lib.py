"""ctypes binding of the C ABI declared in include/tooncrafter_hip.h.

The library is the product: if it is missing this module raises -- there is no
PyTorch or CPU fallback behind it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtooncrafter_hip.so")

TC_ABI_VERSION = 13
ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU = 0, 1, 2, 3
GATHER_LINEAR, GATHER_CONV3x3, GATHER_CONVT3 = 0, 1, 2

ERRORS = {-1: "TC_EINVAL (null pointer / bad size)", -2: "TC_EALIGN (16-byte alignment)",
          -3: "TC_ESHAPE (unsupported shape)", -4: "TC_EWORKSPACE (workspace too small)"}


class TcGemmParams(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p),
        ("bias", C.c_void_p), ("row_bias", C.c_void_p), ("residual", C.c_void_p),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("ldrb", C.c_int32), ("row_div", C.c_int32),
        ("alpha", C.c_float), ("out_scale", C.c_float),
        ("act", C.c_int32), ("out_f32", C.c_int32),
        ("gather", C.c_int32), ("cin", C.c_int32), ("frames", C.c_int32), ("t_len", C.c_int32),
        ("h_out", C.c_int32), ("w_out", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("pad", C.c_int32),
        ("batch", C.c_int32),
        ("stride_a", C.c_int64), ("stride_w", C.c_int64), ("stride_c", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("a_norm", C.c_int32), ("a_norm_eps", C.c_float),
        ("gn_part", C.c_void_p),
    ]


class TcGemmMxParams(C.Structure):
    _fields_ = [("g", TcGemmParams), ("a_scale", C.c_void_p), ("w_scale", C.c_void_p),
                ("lda_s", C.c_int32), ("ldw_s", C.c_int32)]


class TcAttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("batch", C.c_int32), ("heads", C.c_int32), ("lq", C.c_int32), ("lk", C.c_int32),
        ("q_sb", C.c_int64), ("k_sb", C.c_int64), ("v_sb", C.c_int64), ("o_sb", C.c_int64),
        ("q_ss", C.c_int32), ("k_ss", C.c_int32), ("v_ss", C.c_int32), ("o_ss", C.c_int32),
        ("kv_bdiv", C.c_int32), ("accumulate", C.c_int32), ("scale", C.c_float),
        ("k2", C.c_void_p), ("v2", C.c_void_p), ("lk2", C.c_int32), ("kv2_bdiv", C.c_int32),
        ("k2_sb", C.c_int64), ("v2_sb", C.c_int64), ("k2_ss", C.c_int32), ("v2_ss", C.c_int32),
    ]


class TcDdimParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("e_cond", C.c_void_p), ("e_uncond", C.c_void_p), ("noise", C.c_void_p),
        ("x_prev", C.c_void_p), ("pred_x0", C.c_void_p),
        ("b", C.c_int32), ("n", C.c_int64),
        ("cfg_scale", C.c_float), ("guidance_rescale", C.c_float),
        ("sqrt_ac", C.c_float), ("sqrt_1m_ac", C.c_float), ("sqrt_a_prev", C.c_float),
        ("dir_coef", C.c_float), ("sigma", C.c_float), ("x0_rescale", C.c_float),
        ("e_uncond_img", C.c_void_p), ("cfg_img", C.c_float),
    ]


# name -> (restype, argtypes): every symbol include/tooncrafter_hip.h declares
class TcFfParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("out", C.c_void_p),
        ("m", C.c_int32), ("c", C.c_int32), ("hidden", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("ln", C.c_int32),
        ("ln_eps", C.c_float),
    ]


class TcTbParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("wo", C.c_void_p), ("bo", C.c_void_p), ("out", C.c_void_p),
        ("b", C.c_int32), ("t", C.c_int32), ("hw", C.c_int32), ("c", C.c_int32), ("heads", C.c_int32),
        ("ldx", C.c_int32), ("ldo", C.c_int32), ("ln", C.c_int32),
        ("ln_eps", C.c_float), ("scale", C.c_float),
    ]


class TcTqaParams(C.Structure):
    """ABI 13: temporal qkv projection + attention as one launch (csrc/qkv_attn.hip)."""
    _fields_ = [
        ("x", C.c_void_p), ("wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("out", C.c_void_p),
        ("b", C.c_int32), ("t", C.c_int32), ("hw", C.c_int32), ("c", C.c_int32), ("heads", C.c_int32),
        ("ldx", C.c_int32), ("ldo", C.c_int32),
        ("scale", C.c_float),
    ]


TC_PREFETCH_MAX = 4


class TcPrefetch(C.Structure):
    """ABI 12: read-only device buffers (a consumer's weights) that extra blocks of a norm launch stream into the
    Infinity Cache."""
    _fields_ = [("ptr", C.c_void_p * TC_PREFETCH_MAX), ("bytes", C.c_int64 * TC_PREFETCH_MAX), ("n", C.c_int32)]


SYMBOLS = {
    "tc_gemm_bf16": (C.c_int, [C.POINTER(TcGemmParams), C.c_void_p]),
    "tc_gemm_workspace": (C.c_int64, [C.POINTER(TcGemmParams)]),
    "tc_gemm_mxfp8": (C.c_int, [C.POINTER(TcGemmMxParams), C.c_void_p]),
    "tc_quant_mxfp8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                 C.c_void_p]),
    "tc_layernorm_mxfp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "tc_attn_d64": (C.c_int, [C.POINTER(TcAttnParams), C.c_void_p]),
    "tc_attn_temporal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_float, C.c_void_p]),
    "tc_groupnorm_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "tc_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "tc_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                               C.c_void_p]),
    "tc_groupnorm_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(TcPrefetch), C.c_void_p]),
    "tc_layernorm_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                  C.POINTER(TcPrefetch), C.c_void_p]),
    "tc_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "tc_nchw_to_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "tc_rows_to_nchw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "tc_concat_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "tc_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tc_silu_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tc_timestep_embedding_i64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tc_repeat_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "tc_time_mix3": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_int32, C.c_void_p]),
    "tc_video_to_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tc_ddim_workspace": (C.c_int64, [C.c_int32]),
    "tc_ddim_step": (C.c_int, [C.POINTER(TcDdimParams), C.c_void_p, C.c_int64, C.c_void_p]),
    "tc_gemm_ws_eligible": (C.c_int, [C.POINTER(TcGemmParams)]),
    "tc_gemm_gn_rows": (C.c_int, [C.POINTER(TcGemmParams)]),
    "tc_groupnorm_part": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "tc_ff_geglu_fused_eligible": (C.c_int, [C.POINTER(TcFfParams)]),
    "tc_ff_geglu_fused": (C.c_int, [C.POINTER(TcFfParams), C.c_void_p]),
    "tc_temporal_attn_fused_eligible": (C.c_int, [C.POINTER(TcTbParams)]),
    "tc_temporal_attn_fused": (C.c_int, [C.POINTER(TcTbParams), C.c_void_p]),
    "tc_temporal_qkv_attn_eligible": (C.c_int, [C.POINTER(TcTqaParams)]),
    "tc_temporal_qkv_attn": (C.c_int, [C.POINTER(TcTqaParams), C.c_void_p]),
    "tc_abi_version": (C.c_int, []),
    "tc_build_info": (C.c_char_p, []),
}

_lib = None


class TooncrafterHipError(RuntimeError):
    pass


class TooncrafterAbiError(TooncrafterHipError):
    """A library built against another ABI version was loaded: a stale build, never a reason to fall back."""


def load():
    """dlopen the in-tree library and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TooncrafterHipError(
            f"{LIB_PATH} not found: build it with `python -m tooncrafter_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.tc_abi_version()
    if ver != TC_ABI_VERSION:
        raise TooncrafterHipError(f"ABI mismatch: library {ver}, binding {TC_ABI_VERSION}")
    if os.environ.get("TC_DEBUG_SYNC"):
        lib = _TracedLib(lib)
    _lib = lib
    return lib


class _TracedLib:
    """TC_DEBUG_SYNC=1: print every entry-point call (arguments, GEMM/attention parameter blocks) to stderr
    BEFORE it runs and synchronise the device after it, so that a GPU memory fault -- which kills the
    process asynchronously -- is attributable to the last line printed."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in SYMBOLS or name in ("tc_abi_version", "tc_build_info") or name.endswith("_workspace"):
            return fn
        import sys

        import torch

        def show(a):
            s = getattr(a, "_obj", None)
            if isinstance(s, C.Structure):
                vals = []
                for f, _ in s._fields_:
                    v = getattr(s, f)
                    vals.append("%s=%s" % (f, hex(v) if isinstance(v, int) and v > 1 << 20 else v))
                return "{" + " ".join(vals) + "}"
            return hex(a) if isinstance(a, int) and a > 1 << 20 else repr(a)

        def traced(*args):
            sys.stderr.write("[tc] %s %s\n" % (name, " ".join(show(a) for a in args)))
            sys.stderr.flush()
            rc = fn(*args)
            torch.cuda.synchronize()
            return rc
        return traced


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise TooncrafterHipError(f"{what}: {ERRORS.get(rc, rc)}")
    raise TooncrafterHipError(f"{what}: hipError_t {rc}")
