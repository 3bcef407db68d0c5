// Helpers shared by the GEMM kernels (gemm.hip: 128x128 / 64x64 tiles, gemm_wide.hip: 256-row tiles).
#pragma once
#include "common.h"

constexpr int TC_BK = 64;   // K-step of every GEMM kernel: one 128-byte LDS row per tile row

// Byte offset of 16-byte chunk `chunk` (0..7) of tile row `row` in the LDS image.  Rows are 128 B; the
// chunk index is XOR-swizzled by (row>>1)&7, which makes the MFMA fragment reads (16 lanes reading
// the same logical chunk of 16 different rows with ds_read_b128) bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * (TC_BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ u32x4 mask4(const u32x4& v, bool keep) {
  const uint32_t m = keep ? 0xffffffffu : 0u;
  return u32x4{v[0] & m, v[1] & m, v[2] & m, v[3] & m};
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == TC_ACT_SILU) return silu_f(v);
  if (act == TC_ACT_GELU) return gelu_erf_f(v);
  return v;
}

// Per-thread gather state for `R` rows of the A tile (row = lrow + 64*i or lrow + 32*i).
template <int GATHER, int R>
struct AGather {
  bool ok[R];
  int m[R];                 // clamped output row
  int f[R], y[R], x[R];     // frame / y / x (CONV3x3); t-in-clip in y (CONVT3)

  __device__ __forceinline__ void init(const TcGemmParams& p, int tile_row0, int lrow, int row_step) {
    const int hw = p.h_out * p.w_out;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int mm = tile_row0 + lrow + row_step * i;
      ok[i] = mm < p.m;
      const int mc = ok[i] ? mm : p.m - 1;
      m[i] = mc;
      f[i] = y[i] = x[i] = 0;
      if (GATHER == TC_GATHER_CONV3x3) {
        const int q = mc / p.w_out;
        x[i] = mc - q * p.w_out;
        f[i] = q / p.h_out;
        y[i] = q - f[i] * p.h_out;
      } else if (GATHER == TC_GATHER_CONVT3) {
        y[i] = (mc / hw) % p.t_len;
      }
    }
  }

  // Issues the loads of this thread's 16-byte chunk of each of its rows for the K-block starting at
  // k0.  Branch-free: out-of-range rows/taps read a clamped in-bounds address; bit i of the returned
  // mask says whether row i is real.  The caller zeroes invalid rows when it WRITES them to LDS
  // (apply_mask) -- touching the loaded registers here would make the compiler wait for the loads
  // before the MFMAs of the current tile instead of overlapping them.
  __device__ __forceinline__ unsigned load(const TcGemmParams& p, const bf16_t* __restrict__ a_base, int k0, int chunk,
                                           u32x4 (&ra)[R]) const {
    const bool k_ok = k0 + chunk * 8 < p.k;
    const int kc = k_ok ? k0 + chunk * 8 : 0;
    bool v[R];
    if (GATHER == TC_GATHER_LINEAR) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(a_base + (int64_t)m[i] * p.lda + kc);
        v[i] = ok[i];
      }
    } else if (GATHER == TC_GATHER_CONV3x3) {
      const int hv = p.upsample ? p.h_in * 2 : p.h_in;
      const int wv = p.upsample ? p.w_in * 2 : p.w_in;
      const int tap = k0 / p.cin;
      const int c0 = k0 - tap * p.cin + chunk * 8;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        int iy = y[i] * p.stride + dy;
        int ix = x[i] * p.stride + dx;
        v[i] = ok[i] && iy >= 0 && iy < hv && ix >= 0 && ix < wv;
        iy = v[i] ? iy : 0;
        ix = v[i] ? ix : 0;
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        const int64_t src = ((int64_t)f[i] * p.h_in + iy) * p.w_in + ix;
        ra[i] = *reinterpret_cast<const u32x4*>(a_base + src * p.lda + c0);
      }
    } else {  // CONVT3
      const int hw = p.h_out * p.w_out;
      const int tap = k0 / p.cin;
      const int c0 = k0 - tap * p.cin + chunk * 8;
      const int dt = tap - 1;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int tt = y[i] + dt;
        v[i] = ok[i] && tt >= 0 && tt < p.t_len;
        const int64_t src = (int64_t)m[i] + (v[i] ? (int64_t)dt * hw : 0);
        ra[i] = *reinterpret_cast<const u32x4*>(a_base + src * p.lda + c0);
      }
    }
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) mask |= (v[i] && k_ok) ? (1u << i) : 0u;
    return mask;
  }
};

template <int R>
__device__ __forceinline__ void apply_mask(u32x4 (&r)[R], unsigned mask) {
#pragma unroll
  for (int i = 0; i < R; ++i) r[i] = mask4(r[i], (mask >> i) & 1u);
}

int tc_gemm_wide_try(const TcGemmParams& p, int batch, hipStream_t s, bool force);   // gemm_wide.hip; 1 = launched
