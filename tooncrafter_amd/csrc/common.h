// Shared device helpers for the gfx950 kernels. Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tooncrafter_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define TC_WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
  return __uint_as_float(bits16 << 16);
}

// unpack 8 bf16 held in a 16-byte vector to fp32
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)lo;   // round-to-nearest-even (v_cvt_pk_bf16_f32)
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

// SiLU on the hardware's reciprocal and base-2 exponential (v_rcp_f32 / v_exp_f32, 1 ulp each: |error| <= 3e-7 relative,
// four orders of magnitude below a bf16 ulp): `x / (1 + __expf(-x))` compiles to the correctly rounded division -- two
// v_div_scale, v_rcp, six fused multiply-adds, v_div_fmas, v_div_fixup -- and a range-checked exponential, 17 vector
// instructions per element of every GroupNorm + SiLU pass (64 x {2 div_scale, div_fmas, div_fixup} in gn_apply's ISA);
// this form is 5.  Limits as before: x -> -inf gives -0 (exp2 = +inf, rcp = 0), +inf stays +inf, NaN stays NaN.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp): branch-free and an
// order of magnitude less code than libdevice's erff, which matters in the unrolled GEMM epilogue.
// The reciprocal is the hardware's v_rcp_f32 (1 ulp): the correctly rounded one is an eleven-instruction
// division sequence, and a GEGLU epilogue at K = 320 is bound by exactly this arithmetic, not by its MFMAs.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// GELU (erf form, torch.nn.functional.gelu's default) on the same approximation, arranged for the fewest VALU
// instructions (11 full-rate + rcp + exp): with z = x / sqrt 2, t = 1 / (1 + p |z|) and u = P(t) exp(-z^2) / 2 (>= 0),
//     x >= 0:  x (1 + erf z) / 2 = x - |x| u          x < 0:  x (1 - erf |z|) / 2 = -|x| u
// i.e. gelu(x) = max(x, 0) - |x| u.
__device__ __forceinline__ float gelu_erf_f(float x) {
#if defined(TC_GELU_VARIANT) && TC_GELU_VARIANT == 0          /* experiment arms only (scripts/gelu_ab.sh) */
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t0 = __frcp_rn(1.0f + 0.3275911f * az);
  const float p0 = t0 * (0.254829592f + t0 * (-0.284496736f + t0 * (1.421413741f + t0 * (-1.453152027f + t0 * 1.061405429f))));
  return 0.5f * x * (1.0f + copysignf(1.0f - p0 * __expf(-az * az), z));
#elif defined(TC_GELU_VARIANT) && TC_GELU_VARIANT == 2
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
#endif
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float poly = t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f +
                     t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
  const float e = __builtin_amdgcn_exp2f((x * (-0.5f * 1.44269504088896340736f)) * x);
  return __builtin_fmaf(-ax, poly * e, fmaxf(x, 0.0f));
}

// Two at a time: the same arithmetic on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- two lanes of fp32 per
// instruction at the scalar one's rate), bit-identical to gelu_erf_f per element; the reciprocal, the exponential, |x| and
// max(x, 0) stay scalar.  16 + 4 quarter-rate instructions per pair against 2 x (11 + 2).
typedef float tc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tc_f32x2 gelu_erf_f2(tc_f32x2 x) {
  const tc_f32x2 ax = __builtin_elementwise_abs(x);
  const tc_f32x2 d = __builtin_elementwise_fma(ax, (tc_f32x2)(0.3275911f * 0.70710678118654752440f), (tc_f32x2)(1.0f));
  tc_f32x2 t;
  t[0] = __builtin_amdgcn_rcpf(d[0]);
  t[1] = __builtin_amdgcn_rcpf(d[1]);
  const tc_f32x2 poly = t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f +
                        t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
  const tc_f32x2 w = (x * (-0.5f * 1.44269504088896340736f)) * x;
  tc_f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(w[0]);
  e[1] = __builtin_amdgcn_exp2f(w[1]);
  return __builtin_elementwise_fma(-ax, poly * e, __builtin_elementwise_max(x, (tc_f32x2)(0.0f)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

static inline bool tc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define TC_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
