// Epilogues over the fp32 output tile staged in LDS, shared by the bf16 (gemm.hip) and MX-fp8 (gemm_mx.hip)
// 4-wave kernels: bias / row bias / activation / GEGLU / residual on 16-byte row vectors.
#pragma once
#include "gemm_common.h"

// The operands of the epilogue that come from global memory -- bias vectors and the residual rows -- are requested EARLY
// (EpiPrefetch): the bias before the K loop, the residual in front of the last K-step, so their latency runs under the
// MFMAs instead of standing between the K loop and the stores.  With K as short as 320-640 a tile's life is a dependent
// chain (first loads -> K-steps -> LDS transpose -> bias/residual loads -> stores) that two resident blocks per CU only
// half hide; every exposed memory round trip taken out of it is a few per cent of those launches.
template <bool GEGLU, int BM, int BN>
struct EpiGeo {
  static constexpr int GROUPS = GEGLU ? BN / 16 : BN / 8;   // 8-column groups per output row of this tile
  static constexpr int ITERS = BM * GROUPS / 256;
  static constexpr int ROWS_PER_IT = 256 / GROUPS;
};

// GroupNorm statistics of the tile (ABI 9, TcGemmParams.gn_part): per thread the sum and the sum of squares of the
// bf16-ROUNDED outputs of its 8 columns over its rows of the tile; gemm.hip folds them over the tile's rows
struct EpiStats {
  float s[8], q[8];
};

struct EpiPrefetch {
  float bv[8], bg[8];       // bias of this thread's 8 output columns (GEGLU: values | gates)
  u32x4 rres[8];            // residual vectors of its rows (non-GEGLU tiles of at most 128 x 128 / 256 threads)
  bool have_res;
};

// GEGLU weights are packed per 32 rows as [16 values | 16 gates]: output column j of the tile lives at packed column
// 32*(j/16) + j%16, its gate 16 columns further
template <bool GEGLU>
__device__ __forceinline__ int epi_packed_col(int g) { return GEGLU ? 32 * ((g * 8) / 16) + (g * 8) % 16 : g * 8; }

template <bool GEGLU, int BM, int BN>
__device__ __forceinline__ void epi_load_bias(const TcGemmParams& p, int tid, int tile_n, EpiPrefetch& e) {
  using G = EpiGeo<GEGLU, BM, BN>;
  const int g = tid % G::GROUPS;
  const int n_out = GEGLU ? p.n / 2 : p.n;
  const int n0 = (GEGLU ? tile_n * (BN / 2) : tile_n * BN) + g * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { e.bv[i] = 0.f; e.bg[i] = 0.f; }
  e.have_res = false;
  if (n0 >= n_out || !p.bias) return;
  const int pc = epi_packed_col<GEGLU>(g);
  const float* bp = p.bias + (GEGLU ? tile_n * BN + pc : n0);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { e.bv[i] = b0[i]; e.bv[4 + i] = b1[i]; }
  if (GEGLU) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(bp + 16), g1 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
    for (int i = 0; i < 4; ++i) { e.bg[i] = g0[i]; e.bg[4 + i] = g1[i]; }
  }
}

// residual rows of a non-GEGLU tile (GEGLU launches carry none)
template <int BM, int BN>
__device__ __forceinline__ void epi_load_residual(const TcGemmParams& p, int tid, int tile_m, int tile_n, int64_t bz,
                                                  EpiPrefetch& e) {
  using G = EpiGeo<false, BM, BN>;
  static_assert(G::ITERS <= 8, "EpiPrefetch::rres holds 8 vectors");
  e.have_res = true;
  const int g = tid % G::GROUPS, row0 = tid / G::GROUPS;
  const int n0 = tile_n * BN + g * 8;
  const bf16_t* res_base = reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c;
#pragma unroll
  for (int it = 0; it < G::ITERS; ++it) {
    const int m = tile_m * BM + row0 + it * G::ROWS_PER_IT;
    const int mc = m < p.m ? m : p.m - 1;
    e.rres[it] = u32x4{0u, 0u, 0u, 0u};
    if (n0 < p.n) e.rres[it] = *reinterpret_cast<const u32x4*>(res_base + (int64_t)mc * p.ldr + n0);
  }
}

// ---- epilogue over the fp32 tile staged in LDS ----------------------------------------
// Fast path: every 8-column vector of the tile is fully inside N and 16-byte addressable.
// PLAIN = the common "acc + bias (+ residual)" case (alpha = out_scale = 1, no activation, no row bias):
// with K as short as 320 the epilogue is a third of a block's instructions, so it gets its own
// straight-line instance without the per-element multiplies and activation selects.
// `pre`: bias already loaded by epi_load_bias; residual loaded by epi_load_residual if pre.have_res.
template <bool GEGLU, int BM, int BN, bool PLAIN, bool STATS = false>
__device__ __forceinline__ void epilogue_fast(const TcGemmParams& p, const float* cs, int tid, int tile_m, int tile_n,
                                              int64_t bz, EpiPrefetch& pre, EpiStats* st = nullptr) {
  using G = EpiGeo<GEGLU, BM, BN>;
  constexpr int GROUPS = G::GROUPS, ITERS = G::ITERS, ROWS_PER_IT = G::ROWS_PER_IT;
  const int g = tid % GROUPS;
  const int row0 = tid / GROUPS;
  const int n_out = GEGLU ? p.n / 2 : p.n;
  const int n0 = (GEGLU ? tile_n * (BN / 2) : tile_n * BN) + g * 8;
  if (n0 >= n_out) return;
  const int pc = epi_packed_col<GEGLU>(g);
  const float (&bv)[8] = pre.bv;
  const float (&bg)[8] = pre.bg;
  const bool with_res = !GEGLU && p.residual != nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  if (!GEGLU) {
    if (with_res && !pre.have_res) epi_load_residual<BM, BN>(p, tid, tile_m, tile_n, bz, pre);
  }
  // row-bias loads first
  f32x4 rb0[ITERS], rb1[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int m = tile_m * BM + row0 + it * ROWS_PER_IT;
    const int mc = m < p.m ? m : p.m - 1;
    rb0[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    rb1[it] = rb0[it];
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
      for (int e = 0; e < 8; e += 2) {          // pairs: packed fp32 arithmetic (common.h gelu_erf_f2)
#if defined(TC_ABLATE) && (TC_ABLATE & 8)      // scripts/ablate_gemm.sh: the GEGLU epilogue without its erf
        if (PLAIN) { x[e] = (x[e] + bv[e]) * (gt[e] + bg[e]); x[e + 1] = (x[e + 1] + bv[e + 1]) * (gt[e + 1] + bg[e + 1]); }
#else
        if (PLAIN) {
          const tc_f32x2 v = {x[e] + bv[e], x[e + 1] + bv[e + 1]};
          const tc_f32x2 h = v * gelu_erf_f2(tc_f32x2{gt[e] + bg[e], gt[e + 1] + bg[e + 1]});
          x[e] = h[0]; x[e + 1] = h[1];
        }
#endif
        else {
          const tc_f32x2 v = {x[e] * p.alpha + bv[e], x[e + 1] * p.alpha + bv[e + 1]};
          const tc_f32x2 h = v * gelu_erf_f2(tc_f32x2{gt[e] * p.alpha + bg[e], gt[e + 1] * p.alpha + bg[e + 1]}) * p.out_scale;
          x[e] = h[0]; x[e + 1] = h[1];
        }
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
    if (with_res) {
      float rf[8];
      unpack8(pre.rres[it < 8 ? it : 0], rf);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += rf[e];
    }
#if defined(TC_ABLATE) && (TC_ABLATE & 16)     // ... without its global stores
    if (x[0] == 1.2345e30f) {
#else
    if (m < p.m) {
#endif
      if (p.out_f32) {
        float* op = reinterpret_cast<float*>(c_base) + (int64_t)m * p.ldc + n0;
        *reinterpret_cast<f32x4*>(op) = f32x4{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{x[4], x[5], x[6], x[7]};
      } else {
        const u32x4 packed = pack8(x);
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = packed;
        if (STATS) {
          float fr[8];
          unpack8(packed, fr);                     // what GroupNorm will read back: the rounded values
#pragma unroll
          for (int e = 0; e < 8; ++e) { st->s[e] += fr[e]; st->q[e] += fr[e] * fr[e]; }
        }
      }
    }
  }
}

// the same with every operand fetched inside the epilogue (callers without an early-prefetch point)
template <bool GEGLU, int BM, int BN, bool PLAIN>
__device__ __forceinline__ void epilogue_fast(const TcGemmParams& p, const float* cs, int tid, int tile_m, int tile_n,
                                              int64_t bz) {
  EpiPrefetch pre;
  epi_load_bias<GEGLU, BM, BN>(p, tid, tile_n, pre);
  epilogue_fast<GEGLU, BM, BN, PLAIN>(p, cs, tid, tile_m, tile_n, bz, pre);
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
