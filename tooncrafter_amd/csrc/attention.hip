// Attention kernels, head dim 64, gfx950.
//
// attn_d64_kernel: flash-style softmax(q k^T * scale) v.  One block = 128 query rows of one
// (batch, head): 4 waves x 32 rows.  Per 64-key tile the block stages K row-major and V
// TRANSPOSED ([d][key]) in LDS; each wave then computes S^T = K Q^T with
// v_mfma_f32_32x32x16_bf16 ("swapped" product: every lane owns ONE query column, so the
// softmax row reductions are in-lane plus a single lane^32 exchange), exponentiates in
// registers, and feeds the probabilities straight back as the B operand of
// O^T += V^T P^T -- the accumulator-to-operand key permutation is absorbed by reading the
// V^T fragments with the same permutation, so P never touches LDS.
//
// attn_temporal_kernel: 16-frame (or shorter) self-attention at every pixel; the whole
// problem is 16x16x64 per (pixel, head), done on the VALU by one wave.
#include "gemm_common.h"

#include <stdlib.h>

namespace {

// v_exp_f32 directly: exp2f() expands to six instructions per value (denormal-range test, two selects, add, exp,
// ldexp) -- with 32 scores per lane per tile that alone was 200 of the ~900 VALU instructions per tile of a kernel
// the counters show VALU-bound (29 VALU per MFMA, profiles/r02_pmc_attention.json).  Arguments here are <= 0 and
// results below 2^-126 may flush to zero: exactly what a masked or negligible probability should be.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int KT = 64;            // keys per tile
constexpr int K_STRIDE = 144;     // bytes per K row in LDS (128 + 16 pad): conflict-free ds_read_b128
constexpr int VT_STRIDE = 136;    // bytes per V^T row (64 keys * 2 + 8 pad): conflict-free ds_read_b64
constexpr int K_BYTES = KT * K_STRIDE;     // 9216
constexpr int VT_BYTES = 64 * VT_STRIDE;   // 8704

__global__ __launch_bounds__(256) void attn_d64_kernel(const TcAttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[K_BYTES + VT_BYTES];
  char* ks = smem;
  char* vts = smem + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kvb = b / p.kv_bdiv;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_sb + h * 64;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)kvb * p.k_sb + h * 64;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)kvb * p.v_sb + h * 64;
  bf16_t* ob = reinterpret_cast<bf16_t*>(p.o) + (int64_t)b * p.o_sb + h * 64;

  const int q_row = blockIdx.x * 128 + wave * 32 + l31;
  const int q_ld = q_row < p.lq ? q_row : p.lq - 1;   // clamp: tail rows compute garbage, never stored

  // Q fragments: B operand of S^T = K Q^T -> lane holds Q[q][16 kk + 8 half + j]
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    qf[kk] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)q_ld * p.q_ss + kk * 16 + half * 8);

  const float c = p.scale * 1.4426950408889634f;   // softmax in base 2
  float m_run = -1e30f, l_run = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

  // K/V tile staging: 64 keys x 64 d = 512 16-byte chunks per matrix, 2 per thread.  Lane-consecutive
  // keys (idx & 63) make the transposed V^T writes 128-byte contiguous per d-row (conflict-free).
  // The loads of tile kt+1 are issued before the MFMAs of tile kt and land in LDS after them, so the
  // global latency hides under the compute instead of being exposed once per tile.
  u32x4 kreg[2], vreg[2];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_kv = [&](int kt) {
    const int key0 = kt * KT;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      const int key = idx & 63, dch = idx >> 6;
      const int gk = key0 + key;
      const int gkc = gk < p.lk ? gk : p.lk - 1;          // clamped address, zeroed at store time
      kreg[it] = *reinterpret_cast<const u32x4*>(kb + (int64_t)gkc * p.k_ss + dch * 8);
      vreg[it] = *reinterpret_cast<const u32x4*>(vb + (int64_t)gkc * p.v_ss + dch * 8);
    }
  };
  auto store_kv = [&](int kt) {
    const int key0 = kt * KT;
    uint16_t* vt = reinterpret_cast<uint16_t*>(vts);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      const int key = idx & 63, dch = idx >> 6;
      const bool ok = key0 + key < p.lk;
      const u32x4 kv4 = ok ? kreg[it] : zero4;
      const u32x4 vv4 = ok ? vreg[it] : zero4;
      *reinterpret_cast<u32x4*>(ks + key * K_STRIDE + dch * 16) = kv4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vt[(dch * 8 + 2 * e) * (VT_STRIDE / 2) + key] = (uint16_t)(vv4[e] & 0xffffu);
        vt[(dch * 8 + 2 * e + 1) * (VT_STRIDE / 2) + key] = (uint16_t)(vv4[e] >> 16);
      }
    }
  };

  const int n_tiles = (p.lk + KT - 1) / KT;
  load_kv(0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int key0 = kt * KT;
    __syncthreads();   // previous tile fully consumed
    store_kv(kt);
    __syncthreads();
    if (kt + 1 < n_tiles) load_kv(kt + 1);

    // ---- S^T = K Q^T : two 32-key blocks
    f32x16 st[2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kbk][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (kbk * 32 + l31) * K_STRIDE + kk * 32 + half * 16);
        st[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kbk], 0, 0, 0);
      }
    }
    // lane owns query l31; st[kbk][r] is key  key0 + kbk*32 + (r&3) + 8*(r>>2) + 4*half.
    // The softmax is kept lean (it, not the MFMAs, bounds this kernel): masking only on the ragged
    // last tile, the scale folded into one fma per element, and the O/l rescale skipped whenever no
    // row of the wave raised its running maximum (exact: alpha would be 1).
    if (key0 + KT > p.lk) {
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          st[kbk][r] = key < p.lk ? st[kbk][r] : -1e30f;
        }
    }
    float mx = st[0][0];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kbk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c);         // c > 0: max commutes with the scale
    if (!__all(m_new == m_run)) {
      const float alpha = fast_exp2(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = m_new;
    }
    float rs = 0.f;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(fmaf(st[kbk][r], c, -m_run));   // masked keys: exp2(-huge) = 0
        st[kbk][r] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run += rs;

    // ---- O^T += V^T P^T.  MFMA k-slot (half, j) of slab s in key block kbk carries key
    //      kbk*32 + 16 s + 8 (j>>2) + 4 half + (j&3)  -- for BOTH operands.
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)st[kbk][8 * s + j];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const char* vrow = vts + (d * 32 + l31) * VT_STRIDE + (kbk * 32 + 16 * s + 4 * half) * 2;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);        // keys +0..3
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 16);   // keys +8..11
          u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[d], 0, 0, 0);
        }
      }
  }

  // ---- normalise and store.  oacc[d][r] = O[q = l31][dim = d*32 + (r&3) + 8*(r>>2) + 4*half]
  if (q_row < p.lq) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = ob + (int64_t)q_row * p.o_ss;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dim = d * 32 + 8 * g + 4 * half;
        float x0 = oacc[d][4 * g + 0] * inv, x1 = oacc[d][4 * g + 1] * inv;
        float x2 = oacc[d][4 * g + 2] * inv, x3 = oacc[d][4 * g + 3] * inv;
        u32x2* dst = reinterpret_cast<u32x2*>(orow + dim);
        if (p.accumulate) {
          const u32x2 old = *dst;
          x0 += __uint_as_float(old[0] << 16);
          x1 += __uint_as_float(old[0] & 0xffff0000u);
          x2 += __uint_as_float(old[1] << 16);
          x3 += __uint_as_float(old[1] & 0xffff0000u);
        }
        u32x2 out = {pack2(x0, x1), pack2(x2, x3)};
        *dst = out;
      }
  }
}

// ---------------------------------------------------------------------------------------
// attn_d64_dma_kernel: same mathematics and register-level dataflow as attn_d64_kernel (swapped S^T = K Q^T, in-lane
// softmax, P fed back as the B operand of O^T += V^T P^T), different STAGING:
//   * K and V tiles travel global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB pieces, out-of-range keys
//     zero-filled by the buffer bounds check) into a double-buffered image: the loads of tile t+1 are issued right
//     after the single barrier of tile t and land under its MFMAs -- no staging registers, no ds_write pass (the
//     register path spent 18 LDS stores per thread per tile, 16 of them 2-byte transposing stores), one barrier
//     per tile instead of two;
//   * V stays ROW-major [key][d] in LDS and the V^T fragments of the PV product are read with
//     ds_read_b64_tr_b16 (hardware 4x4 transpose: a 16-lane group reads a [4 keys][16 d] block, every lane
//     gets the 4 keys of its own d column);
//   * both images are XOR-swizzled on the SOURCE side of the DMA (the LDS destination of a DMA piece is
//     lane-linear): K chunks by (row>>1)&7 (conflict-free ds_read_b128 of 16 different rows), V chunks by
//     ((row>>1)&1)<<2 (the four key rows of a transposing read land in four different bank quarters).
constexpr int KD_TILE = KT * 128;            // 8 KiB: 64 keys x 64 d bf16, 128-byte rows
constexpr int DMA_STAGE = 2 * KD_TILE;       // K + V of one tile

__device__ __forceinline__ u32x2 lds_read_tr16_b64(const char* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}

// (A three-stage K/V ring with two tiles of LDS-DMA in flight -- counted vmcnt, raw barrier, 48 KiB -- was built, bit-identical,
// and measured 0.95-1.00x on every shape, profiles/r03_attn_ring_of_three_ab.txt: four resident blocks per CU already
// hide the tile latency and the third stage costs one of them.  Removed.)
template <bool DUAL>
__global__ __launch_bounds__(256) void attn_d64_dma_kernel(const TcAttnParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * DMA_STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_sb + h * 64;
  bf16_t* ob = reinterpret_cast<bf16_t*>(p.o) + (int64_t)b * p.o_sb + h * 64;

  const int q_row = blockIdx.x * 128 + wave * 32 + l31;
  const int q_ld = q_row < p.lq ? q_row : p.lq - 1;   // clamp: tail rows compute garbage, never stored
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    qf[kk] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)q_ld * p.q_ss + kk * 16 + half * 8);

  // ---- fragment read offsets (per lane, fixed for the whole kernel)
  // K (A operand of S^T = K Q^T): lane (key l31 of block kbk, k = 16 kk + 8 half ..): 16-byte chunk 2 kk + half of its row
  int k_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) k_off[kk] = l31 * 128 + (((kk * 2 + half) ^ ((l31 >> 1) & 7)) << 4);
  // V (A operand of O^T += V^T P^T through the transposing read): 16-lane group g = lane>>4 serves the d columns
  // 16 (g&1) .. +15 of d-block d0 with key sub-block 4 half; lane i of the group addresses row i>>2, d 4 (i&3) .. +3
  const int lg = lane & 15;
  int v_off[2];
#pragma unroll
  for (int d0 = 0; d0 < 2; ++d0) {
    const int vrow = lg >> 2;                                   // + key0 (multiple of 4): swizzle bit = (vrow>>1)&1
    const int col = d0 * 32 + 16 * ((lane >> 4) & 1) + 4 * (lg & 3);
    const int chunk = (col >> 3) ^ (((vrow >> 1) & 1) << 2);
    v_off[d0] = vrow * 128 + chunk * 16 + (col & 7) * 2;
  }

  const float c = p.scale * 1.4426950408889634f;   // softmax in base 2
  float m_run = -1e30f, l_run = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  const int prow = lane >> 3;

  // One key/value stream: online softmax over its lk keys into (m_run, l_run, oacc).
  auto run_stream = [&](const bf16_t* kb, const bf16_t* vb, const int lk, const int k_ss, const int v_ss) {
    // descriptors over exactly the lk rows of this (batch, head): a key row >= lk is out of range and reads as zeros
    const tc_rsrc_t k_rsrc = make_rsrc(kb, ((int64_t)(lk - 1) * k_ss + 64) * 2);
    const tc_rsrc_t v_rsrc = make_rsrc(vb, ((int64_t)(lk - 1) * v_ss + 64) * 2);
    // DMA geometry: a tile is 8 pieces of 8 key rows per matrix; wave w issues pieces w and w + 4.  Lane l of a piece
    // lands at (row l>>3, physical chunk l&7) and therefore FETCHES the logical chunk whose swizzled home that is.
    uint32_t k_voff[2], v_voff[2];
    int piece_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 4 * i) * 8 + prow;                  // key row inside the tile
      piece_row[i] = row;
      const int kchunk = (lane & 7) ^ ((row >> 1) & 7);
      const int vchunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
      k_voff[i] = (uint32_t)(row * k_ss * 2 + kchunk * 16);
      v_voff[i] = (uint32_t)(row * v_ss * 2 + vchunk * 16);
    }
    auto dma_tile = [&](int kt, int stage) {
      const int key0 = kt * KT;
      char* sk = smem + stage * DMA_STAGE + wave_u * 1024;
      char* sv = sk + KD_TILE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t dead = (key0 + piece_row[i] < lk) ? 0u : TC_OOB;      // ragged last tile: zero rows
        glds16(k_rsrc, sk + i * 4096, k_voff[i] | dead, (uint32_t)key0 * (uint32_t)k_ss * 2u);
        glds16(v_rsrc, sv + i * 4096, v_voff[i] | dead, (uint32_t)key0 * (uint32_t)v_ss * 2u);
      }
    };

    const int n_tiles = (lk + KT - 1) / KT;
    dma_tile(0, 0);
    for (int kt = 0; kt < n_tiles; ++kt) {
      const int key0 = kt * KT;
      // tile kt has landed for this wave (vmcnt) and for every wave (barrier); the same barrier retires all reads
      // of the buffer the next DMA overwrites
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < n_tiles) dma_tile(kt + 1, (kt + 1) & 1);
      const char* ks = smem + (kt & 1) * DMA_STAGE;
      const char* vs = ks + KD_TILE;

      f32x16 st[2];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kbk * 32 * 128 + k_off[kk]);
          // the first product takes the inline constant 0 as its C operand: no 32 v_mov per tile to clear st
          st[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? zero16 : st[kbk], 0, 0, 0);
        }
      }
      if (key0 + KT > lk) {
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            st[kbk][r] = key < lk ? st[kbk][r] : -1e30f;
          }
      }
      float mx = st[0][0];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kbk][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx * c);
      if (!__all(m_new == m_run)) {
        const float alpha = fast_exp2(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        m_run = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(fmaf(st[kbk][r], c, -m_run));
          st[kbk][r] = pv;
          rs += pv;
        }
      rs += __shfl_xor(rs, 32, 64);
      l_run += rs;

      // O^T += V^T P^T: k-slot (half, j) of slab s in key block kbk carries key kbk*32 + 16 s + 8 (j>>2) + 4 half + (j&3)
      // for both operands; the V side is two transposing reads (keys +0..3 and +8..11 of the slab) per fragment
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)st[kbk][8 * s + j];
          const int kbase = (kbk * 32 + 16 * s + 4 * half) * 128;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const u32x2 lo = lds_read_tr16_b64(vs + kbase + v_off[d]);
            const u32x2 hi = lds_read_tr16_b64(vs + kbase + 8 * 128 + v_off[d]);
            u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[d], 0, 0, 0);
          }
        }
    }
  };

  {
    const int kvb = b / p.kv_bdiv;
    run_stream(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)kvb * p.k_sb + h * 64,
               reinterpret_cast<const bf16_t*>(p.v) + (int64_t)kvb * p.v_sb + h * 64, p.lk, p.k_ss, p.v_ss);
  }
  // second key/value set (text + image cross-attention): normalise the first result, keep it in fp32, start a fresh
  // softmax; the sum is rounded to bf16 once
  f32x16 o1[2];
  if (DUAL) {
    const float inv1 = 1.0f / l_run;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) { o1[d][r] = oacc[d][r] * inv1; oacc[d][r] = 0.f; }
    m_run = -1e30f;
    l_run = 0.f;
    __syncthreads();                     // every wave has finished reading the first stream's last tile
    const int kvb2 = b / p.kv2_bdiv;
    run_stream(reinterpret_cast<const bf16_t*>(p.k2) + (int64_t)kvb2 * p.k2_sb + h * 64,
               reinterpret_cast<const bf16_t*>(p.v2) + (int64_t)kvb2 * p.v2_sb + h * 64, p.lk2, p.k2_ss, p.v2_ss);
  }

  if (q_row < p.lq) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = ob + (int64_t)q_row * p.o_ss;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dim = d * 32 + 8 * g + 4 * half;
        float x0 = oacc[d][4 * g + 0] * inv, x1 = oacc[d][4 * g + 1] * inv;
        float x2 = oacc[d][4 * g + 2] * inv, x3 = oacc[d][4 * g + 3] * inv;
        if (DUAL) { x0 += o1[d][4 * g + 0]; x1 += o1[d][4 * g + 1]; x2 += o1[d][4 * g + 2]; x3 += o1[d][4 * g + 3]; }
        u32x2* dst = reinterpret_cast<u32x2*>(orow + dim);
        if (p.accumulate) {
          const u32x2 old = *dst;
          x0 += __uint_as_float(old[0] << 16);
          x1 += __uint_as_float(old[0] & 0xffff0000u);
          x2 += __uint_as_float(old[1] << 16);
          x3 += __uint_as_float(old[1] & 0xffff0000u);
        }
        u32x2 out = {pack2(x0, x1), pack2(x2, x3)};
        *dst = out;
      }
  }
}

// ---------------------------------------------------------------------------------------
// Temporal attention: one wave per (batch b, pixel p, head h).  Lane = (query i = lane/4,
// quarter = lane%4 owning 16 of the 64 dims).  K and V of the sequence sit in LDS as fp32.
__global__ __launch_bounds__(256) void attn_temporal_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           int nb, int t_len, int hw, int heads, float scale) {
  __shared__ float kv_s[4][2][16][64 + 4];   // [wave][k|v][frame][dim], +4 floats pad
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t seq = (int64_t)blockIdx.x * 4 + wave;       // over nb*hw*heads
  const int64_t total = (int64_t)nb * hw * heads;
  const bool active = seq < total;
  const int64_t sq = active ? seq : total - 1;
  const int hd = (int)(sq % heads);
  const int64_t bp = sq / heads;
  const int px = (int)(bp % hw);
  const int bb = (int)(bp / hw);
  const int C = heads * 64;
  const int ld = 3 * C;
  const int64_t row0 = ((int64_t)bb * t_len) * hw + px;     // frame t -> row0 + t*hw

  // stage K, V: 16 frames x 64 dims each = 128 chunks of 8 dims per matrix; lane handles 2+2
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = lane + it * 64;      // 0..127
    const int fr = idx >> 3, ch = idx & 7;
    float kf[8], vf[8];
    if (fr < t_len) {
      const bf16_t* rp = qkv + (row0 + (int64_t)fr * hw) * ld + hd * 64 + ch * 8;
      unpack8(*reinterpret_cast<const u32x4*>(rp + C), kf);
      unpack8(*reinterpret_cast<const u32x4*>(rp + 2 * C), vf);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { kf[e] = 0.f; vf[e] = 0.f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      kv_s[wave][0][fr][ch * 8 + e] = kf[e];
      kv_s[wave][1][fr][ch * 8 + e] = vf[e];
    }
  }
  const int qi = lane >> 2, quarter = lane & 3;
  float qv[16];
  {
    const int fr = qi < t_len ? qi : t_len - 1;
    const bf16_t* rp = qkv + (row0 + (int64_t)fr * hw) * ld + hd * 64 + quarter * 16;
    unpack8(*reinterpret_cast<const u32x4*>(rp), qv);
    unpack8(*reinterpret_cast<const u32x4*>(rp + 8), qv + 8);
  }
  __syncthreads();

  float s[16];
  float mx = -1e30f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) a += qv[e] * kv_s[wave][0][j][quarter * 16 + e];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a = j < t_len ? a * scale : -1e30f;
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s[j] = __expf(s[j] - mx);
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  float o[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float pj = s[j] * inv;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] += pj * kv_s[wave][1][j][quarter * 16 + e];
  }
  if (active && qi < t_len) {
    bf16_t* op = out + (row0 + (int64_t)qi * hw) * C + hd * 64 + quarter * 16;
    *reinterpret_cast<u32x4*>(op) = pack8(o);
    *reinterpret_cast<u32x4*>(op + 8) = pack8(o + 8);
  }
}

}  // namespace

extern "C" int tc_attn_d64(const TcAttnParams* pp, void* stream) {
  if (!pp) return TC_EINVAL;
  const TcAttnParams& p = *pp;
  if (!p.q || !p.k || !p.v || !p.o) return TC_EINVAL;
  if (p.batch <= 0 || p.heads <= 0 || p.lq <= 0 || p.lk <= 0 || p.kv_bdiv <= 0) return TC_EINVAL;
  if (!tc_aligned16(p.q) || !tc_aligned16(p.k) || !tc_aligned16(p.v) || !tc_aligned16(p.o)) return TC_EALIGN;
  if ((p.q_ss & 7) || (p.k_ss & 7) || (p.v_ss & 7) || (p.o_ss & 7)) return TC_EALIGN;
  if ((p.q_sb & 7) || (p.k_sb & 7) || (p.v_sb & 7) || (p.o_sb & 7)) return TC_EALIGN;
  if (p.heads > 65535 || p.batch > 65535) return TC_ESHAPE;
  const bool dual = p.k2 != nullptr;
  if (dual) {
    if (!p.v2 || p.lk2 <= 0 || p.kv2_bdiv <= 0) return TC_EINVAL;
    if (!tc_aligned16(p.k2) || !tc_aligned16(p.v2) || (p.k2_ss & 7) || (p.v2_ss & 7) || (p.k2_sb & 7) || (p.v2_sb & 7))
      return TC_EALIGN;
  }
  dim3 grid((p.lq + 127) / 128, p.heads, p.batch), block(256);
  // TC_ATTN_STAGE=reg selects the register-staged kernel (A/B runs); the DMA-staged one needs 31-bit row offsets
  static const bool use_reg = [] { const char* e = getenv("TC_ATTN_STAGE"); return e && e[0] == 'r'; }();
  const bool fits = (int64_t)p.lk * p.k_ss * 2 < 0x7fffff00LL && (int64_t)p.lk * p.v_ss * 2 < 0x7fffff00LL &&
                    p.k_ss >= 64 && p.v_ss >= 64;
  if (dual) {
    const bool fits2 = (int64_t)p.lk2 * p.k2_ss * 2 < 0x7fffff00LL && (int64_t)p.lk2 * p.v2_ss * 2 < 0x7fffff00LL &&
                       p.k2_ss >= 64 && p.v2_ss >= 64;
    if (!fits || !fits2) return TC_ESHAPE;
    hipLaunchKernelGGL(attn_d64_dma_kernel<true>, grid, block, 0, reinterpret_cast<hipStream_t>(stream), p);
  } else if (!use_reg && fits) {
    hipLaunchKernelGGL(attn_d64_dma_kernel<false>, grid, block, 0, reinterpret_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(attn_d64_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), p);
  }
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_attn_temporal(const tc_bf16* qkv, tc_bf16* out, int32_t b, int32_t t, int32_t hw,
                                int32_t heads, float scale, void* stream) {
  if (!qkv || !out || b <= 0 || t <= 0 || hw <= 0 || heads <= 0) return TC_EINVAL;
  if (t > 16) return TC_ESHAPE;
  if (!tc_aligned16(qkv) || !tc_aligned16(out)) return TC_EALIGN;
  const int64_t total = (int64_t)b * hw * heads;
  const int64_t nblk = (total + 3) / 4;
  if (nblk > 0x7fffffffLL) return TC_ESHAPE;
  hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(out), b, t, hw, heads, scale);
  TC_LAUNCH_CHECK();
  return TC_OK;
}
