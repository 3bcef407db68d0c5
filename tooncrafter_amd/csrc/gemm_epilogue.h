// Epilogues over the fp32 output tile staged in LDS, shared by the bf16 (gemm.hip) and MX-fp8 (gemm_mx.hip)
// 4-wave kernels: bias / row bias / activation / GEGLU / residual on 16-byte row vectors.
#pragma once
#include "gemm_common.h"

// ---- epilogue over the fp32 tile staged in LDS ----------------------------------------
// Fast path: every 8-column vector of the tile is fully inside N and 16-byte addressable.
// PLAIN = the common "acc + bias (+ residual)" case (alpha = out_scale = 1, no activation, no row bias):
// with K as short as 320 the epilogue is a third of a block's instructions, so it gets its own
// straight-line instance without the per-element multiplies and activation selects.
template <bool GEGLU, int BM, int BN, bool PLAIN>
__device__ __forceinline__ void epilogue_fast(const TcGemmParams& p, const float* cs, int tid, int tile_m, int tile_n,
                                              int64_t bz) {
  constexpr int GROUPS = GEGLU ? BN / 16 : BN / 8;   // 8-column groups per output row of this tile
  constexpr int ITERS = BM * GROUPS / 256;
  constexpr int ROWS_PER_IT = 256 / GROUPS;
  const int g = tid % GROUPS;
  const int row0 = tid / GROUPS;
  const int n_out = GEGLU ? p.n / 2 : p.n;
  const int n0 = (GEGLU ? tile_n * (BN / 2) : tile_n * BN) + g * 8;
  if (n0 >= n_out) return;
  float bv[8], bg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bv[e] = 0.f; bg[e] = 0.f; }
  // GEGLU weights are packed per 32 rows as [16 values | 16 gates]: output column j of this tile
  // lives at packed column 32*(j/16) + j%16, its gate 16 columns further
  const int pc = GEGLU ? 32 * ((g * 8) / 16) + (g * 8) % 16 : g * 8;
  if (p.bias) {
    const float* bp = p.bias + (GEGLU ? tile_n * BN + pc : n0);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
    if (GEGLU) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(bp + 16), g1 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bg[e] = g0[e]; bg[4 + e] = g1[e]; }
    }
  }
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  // issue all residual / row-bias loads first
  u32x4 rres[ITERS];
  f32x4 rb0[ITERS], rb1[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int m = tile_m * BM + row0 + it * ROWS_PER_IT;
    const int mc = m < p.m ? m : p.m - 1;
    rres[it] = u32x4{0u, 0u, 0u, 0u};
    rb0[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    rb1[it] = rb0[it];
    if (res_base) rres[it] = *reinterpret_cast<const u32x4*>(res_base + (int64_t)mc * p.ldr + n0);
    if (!GEGLU && !PLAIN && p.row_bias) {
      const float* rp = p.row_bias + (int64_t)(mc / p.row_div) * p.ldrb + n0;
      rb0[it] = *reinterpret_cast<const f32x4*>(rp);
      rb1[it] = *reinterpret_cast<const f32x4*>(rp + 4);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int row = row0 + it * ROWS_PER_IT;
    const int m = tile_m * BM + row;
    float x[8];
    {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(cs + row * BN + pc);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(cs + row * BN + pc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
    }
    if (GEGLU) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(cs + row * BN + pc + 16);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(cs + row * BN + pc + 20);
      float gt[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { gt[e] = lo[e]; gt[4 + e] = hi[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (PLAIN) x[e] = (x[e] + bv[e]) * gelu_erf_f(gt[e] + bg[e]);
        else x[e] = (x[e] * p.alpha + bv[e]) * gelu_erf_f(gt[e] * p.alpha + bg[e]) * p.out_scale;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (PLAIN) {
          x[e] += bv[e];
        } else {
          const float rbv = e < 4 ? rb0[it][e] : rb1[it][e - 4];
          x[e] = apply_act(x[e] * p.alpha + bv[e] + rbv, p.act) * p.out_scale;
        }
      }
    }
    if (res_base) {
      float rf[8];
      unpack8(rres[it], rf);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += rf[e];
    }
    if (m < p.m) {
      if (p.out_f32) {
        float* op = reinterpret_cast<float*>(c_base) + (int64_t)m * p.ldc + n0;
        *reinterpret_cast<f32x4*>(op) = f32x4{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{x[4], x[5], x[6], x[7]};
      } else {
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = pack8(x);
      }
    }
  }
}

// Slow path: N not a multiple of 8 (the 4-channel UNet output, the 3-channel decoder output).
template <int BM, int BN>
__device__ __forceinline__ void epilogue_tail(const TcGemmParams& p, const float* cs, int tid, int tile_m, int tile_n,
                                              int64_t bz) {
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  for (int v = tid; v < BM * BN; v += 256) {
    const int row = v / BN, col = v - row * BN;
    const int m = tile_m * BM + row, n = tile_n * BN + col;
    if (m >= p.m || n >= p.n) continue;
    float val = cs[row * BN + col] * p.alpha;
    if (p.bias) val += p.bias[n];
    if (p.row_bias) val += p.row_bias[(int64_t)(m / p.row_div) * p.ldrb + n];
    val = apply_act(val, p.act) * p.out_scale;
    if (res_base) val += (float)res_base[(int64_t)m * p.ldr + n];
    if (p.out_f32) reinterpret_cast<float*>(c_base)[(int64_t)m * p.ldc + n] = val;
    else reinterpret_cast<bf16_t*>(c_base)[(int64_t)m * p.ldc + n] = (bf16_t)val;
  }
}
