// MX block-scaled fp8 GEMM with the same implicit-GEMM row gather and fused epilogue as gemm.hip, gfx950
// (BASELINE.json configs[4]: the CDNA4 fp8 MFMA GEMM path), and the bf16 -> MXFP8 quantiser that feeds it.
//
//   C[M, N] = epilogue( gather(A_q * 2^sa)[M, K] * (W_q * 2^sw)[N, K]^T )
//
// Operands: OCP e4m3 bytes with one E8M0 scale per 32 consecutive K elements of a row; the products and the
// sum are exact in the matrix pipe's fp32 accumulator, the block scales are applied by the instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 MFMA rate).  Measured lane layout of that instruction
// (scripts/probes/mx_probe.hip): lane (l31, h) carries matrix row l31; its bytes 0..15 belong to the FIRST
// 32-element K block of the instruction's 64 and bytes 16..31 to the SECOND, the two halves h = 0/1 splitting
// each block's 32 bytes between them; the lane's scale byte applies to K block h of row l31.
//
// Tiling: the 128x128 (64x64 for the low-resolution layers) 4-wave tile of gemm.hip with a K-step of 128
// ELEMENTS -- the same 128-byte LDS rows, the same global -> LDS DMA (1 KiB = 8 tile rows per wave instruction),
// the same (row>>1)&7 source-side chunk swizzle, so every ds_read_b128 of a fragment is conflict-free -- i.e.
// half the LDS / L1 bytes per FLOP of the bf16 kernel, whose K loop those bytes bound.  One K-step is two MFMA
// K-slices of 64; a lane reads chunks (4 j + h, 4 j + 2 + h) of its row for slice j.  A K-step of a convolution
// may straddle two taps (cin = 320, 960 are not multiples of 128): the tap is decoded PER LANE from the lane's
// own chunk (one v_mul_hi), so any cin % 64 == 0 works.  Scales go global -> registers (2 bytes per row and
// K-slice), issued with the tile loads of the step they belong to.
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BKE = 128;   // K-step in elements = bytes: one 128-byte LDS row per tile row
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Source-row byte offsets of R output rows of the tile for a given tap, for a matrix with leading dimension
// `ld` bytes (the fp8 activations, or their scale matrix): TC_OOB where the row / tap does not exist.
template <int GATHER, int R>
struct MxRows {
  uint32_t base[R];    // byte offset of the centre-tap source row; TC_OOB (linear) if row >= M
  uint32_t vbits[R];   // convolution: bit t set if tap t of this row is inside the image
  int f[R], y[R], x[R];
  bool ok[R];

  __device__ __forceinline__ void init(const TcGemmParams& p, int ld, int row0, int row_step) {
    const int hw = p.h_out * p.w_out;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int mm = row0 + row_step * i;
      ok[i] = mm < p.m;
      const int mc = ok[i] ? mm : 0;
      f[i] = y[i] = x[i] = 0;
      vbits[i] = 0u;
      if (GATHER == TC_GATHER_LINEAR) {
        base[i] = ok[i] ? (uint32_t)((int64_t)mc * ld) : TC_OOB;
      } else if (GATHER == TC_GATHER_CONV3x3) {
        const int q = mc / p.w_out;
        x[i] = mc - q * p.w_out;
        f[i] = q / p.h_out;
        y[i] = q - f[i] * p.h_out;
        base[i] = (uint32_t)((((int64_t)f[i] * p.h_in + y[i]) * p.w_in + x[i]) * ld);
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y[i] + t / 3 - 1, ix = x[i] + t % 3 - 1;
          if (ok[i] && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) bits |= 1u << t;
        }
        vbits[i] = bits;
      } else {  // CONVT3
        base[i] = (uint32_t)((int64_t)mc * ld);
        const int tt = (mc / hw) % p.t_len;
        vbits[i] = ok[i] ? ((tt > 0 ? 1u : 0u) | 2u | (tt + 1 < p.t_len ? 4u : 0u)) : 0u;
      }
    }
  }

  // offsets of tap `tap` (may differ per lane; taps past the last one give TC_OOB) plus `add` bytes into the row
  __device__ __forceinline__ void offsets(const TcGemmParams& p, int ld, int tap, uint32_t add, uint32_t (&voff)[R]) const {
    if (GATHER == TC_GATHER_LINEAR) {
#pragma unroll
      for (int i = 0; i < R; ++i) voff[i] = base[i] + add;          // TC_OOB + add (< 2 GiB) stays out of range
    } else if (GATHER == TC_GATHER_CONV3x3) {
      const int ty = (tap * 11) >> 5;                                // tap / 3 for tap < 16
      const int dy = ty - p.pad, dx = tap - ty * 3 - p.pad;
      if (p.stride == 1 && !p.upsample && p.pad == 1) {
        const uint32_t delta = (uint32_t)((dy * p.w_in + dx) * ld) + add;
#pragma unroll
        for (int i = 0; i < R; ++i) voff[i] = ((vbits[i] >> tap) & 1u) ? base[i] + delta : TC_OOB;
      } else {
        const int hv = p.upsample ? p.h_in * 2 : p.h_in;
        const int wv = p.upsample ? p.w_in * 2 : p.w_in;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          int iy = y[i] * p.stride + dy;
          int ix = x[i] * p.stride + dx;
          const bool v = ok[i] && tap < 9 && iy >= 0 && iy < hv && ix >= 0 && ix < wv;
          if (p.upsample) { iy >>= 1; ix >>= 1; }
          const int64_t src = ((int64_t)f[i] * p.h_in + iy) * p.w_in + ix;
          voff[i] = v ? (uint32_t)(src * ld) + add : TC_OOB;
        }
      }
    } else {  // CONVT3
      const uint32_t delta = (uint32_t)((tap - 1) * p.h_out * p.w_out * ld) + add;
#pragma unroll
      for (int i = 0; i < R; ++i) voff[i] = ((vbits[i] >> tap) & 1u) ? base[i] + delta : TC_OOB;
    }
  }
};

__device__ __forceinline__ uint32_t buf_load_u16(tc_rsrc_t rsrc, uint32_t voff) {
  return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, 0, 0);
}
__device__ __forceinline__ uint32_t buf_load_u32(tc_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0);
}

template <int GATHER, int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_mx_kernel(const TcGemmMxParams px, const int order) {
  const TcGemmParams& p = px.g;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int STAGE_BYTES = (BM + BN) * BKE;                           // 32 KiB per stage at 128x128
  constexpr int EPI_BYTES = BM * BN * 4;
  constexpr int SMEM_BYTES = 2 * STAGE_BYTES > EPI_BYTES ? 2 * STAGE_BYTES : EPI_BYTES;
  constexpr int RA = BM / 32, RB = BN / 32;                              // loader rows per thread
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];      // reused by the epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  int tile_m, tile_n;
  tc_tile_of_block(blockIdx.x, tiles_m, tiles_n, order, tile_m, tile_n);
  if (tile_m >= tiles_m) return;

  const int kc = GATHER == TC_GATHER_LINEAR ? p.k : p.cin;               // K elements per source row
  const int64_t a_rows = tc_a_rows(p);
  const tc_rsrc_t a_rsrc = make_rsrc(p.a, (a_rows - 1) * p.lda + kc);
  const tc_rsrc_t w_rsrc = make_rsrc(p.w, (int64_t)(p.n - 1) * p.ldw + p.k);
  const tc_rsrc_t as_rsrc = make_rsrc(px.a_scale, a_rows * px.lda_s);
  const tc_rsrc_t ws_rsrc = make_rsrc(px.w_scale, (int64_t)p.n * px.ldw_s);
  // tap = kk / cin by one multiply-high: exact while kk * cin < 2^32 (K <= 23040, cin <= 2560)
  const uint32_t magic = GATHER == TC_GATHER_LINEAR ? 0u : 0xffffffffu / (uint32_t)p.cin + 1u;

  // ---- loader geometry: thread -> (row lrow + 32 i, 16-byte chunk) of both tiles, as in gemm.hip
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  MxRows<GATHER, RA> ag;
  ag.init(p, p.lda, tile_m * BM + lrow, 32);
  uint32_t b_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = tile_n * BN + lrow + 32 * i;
    b_voff[i] = n < p.n ? (uint32_t)((int64_t)n * p.ldw + chunk * 16) : TC_OOB;
  }

  auto load_tile = [&](int kb, int stage) {
    const int k0 = kb * BKE;
    const uint32_t kk = (uint32_t)(k0 + chunk * 16);                     // first K element of this lane's chunk
    const uint32_t kill = kk >= (uint32_t)p.k ? TC_OOB : 0u;             // K tail: zero-filled
    uint32_t a_voff[RA];
    if (GATHER == TC_GATHER_LINEAR) {
      ag.offsets(p, p.lda, 0, kk, a_voff);
    } else {
      const int tap = (int)__umulhi(kk, magic);
      ag.offsets(p, p.lda, tap, kk - (uint32_t)(tap * p.cin), a_voff);
    }
    char* sa = smem + stage * STAGE_BYTES + wave_u * 1024;
    char* sb = sa + BM * BKE;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      glds16(w_rsrc, sb + i * 4096, b_voff[i] | kill, (uint32_t)k0);
#pragma unroll
    for (int i = 0; i < RA; ++i)
      glds16(a_rsrc, sa + i * 4096, a_voff[i] | kill, 0u);
  };

  // ---- scales of the fragment rows this lane feeds to the matrix pipe
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  MxRows<GATHER, TM> sg;
  sg.init(p, px.lda_s, tile_m * BM + wm * 32 * TM + frow, 32);
  uint32_t ws_voff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = tile_n * BN + wn * 32 * TN + j * 32 + frow;
    ws_voff[j] = n < p.n ? (uint32_t)((int64_t)n * px.ldw_s) : TC_OOB;
  }
  // sa[s][i]: the two scale bytes (K blocks 2 s, 2 s + 1 of the step) of A fragment row i; sw[j]: all four of W row j
  auto load_scales = [&](int kb, uint32_t (&sa)[2][TM], uint32_t (&sw)[TN]) {
    const int k0 = kb * BKE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t kk = (uint32_t)(k0 + 64 * s);                       // block-uniform: scalar tap decode
      uint32_t voff[TM];
      if (GATHER == TC_GATHER_LINEAR) {
        sg.offsets(p, px.lda_s, 0, kk >> 5, voff);
      } else {
        const int tap = (int)__umulhi(kk, magic);
        sg.offsets(p, px.lda_s, tap, (kk - (uint32_t)(tap * p.cin)) >> 5, voff);
      }
      const uint32_t kill = kk >= (uint32_t)p.k ? TC_OOB : 0u;
#pragma unroll
      for (int i = 0; i < TM; ++i) sa[s][i] = buf_load_u16(as_rsrc, voff[i] | kill);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) sw[j] = buf_load_u32(ws_rsrc, ws_voff[j], (uint32_t)(k0 >> 5));
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage, const uint32_t (&sa)[2][TM], const uint32_t (&sw)[TN]) {
    const char* la = smem + stage * STAGE_BYTES;
    const char* lb = la + BM * BKE;
    i32x8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * 32 * TM + i * 32 + frow;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(la + lds_off(row, 4 * s + fhalf));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(la + lds_off(row, 4 * s + 2 + fhalf));
        af[s][i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * 32 * TN + j * 32 + frow;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(lb + lds_off(row, 4 * s + fhalf));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(lb + lds_off(row, 4 * s + 2 + fhalf));
        bf[s][j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int sca[TM], scb[TN];                     // byte 0 = this lane's scale: K block 2 s + fhalf of the step
#pragma unroll
      for (int i = 0; i < TM; ++i) sca[i] = (int)((sa[s][i] >> (8 * fhalf)) & 0xffu);
#pragma unroll
      for (int j = 0; j < TN; ++j) scb[j] = (int)((sw[j] >> (8 * (2 * s + fhalf))) & 0xffu);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[s][i], bf[s][j], acc[i][j], 0, 0, 0, sca[i], 0, scb[j]);
    }
  };

  const int nk = (p.k + BKE - 1) / BKE;
  uint32_t sa[2][2][TM], sw[2][TN];
  load_tile(0, 0);
  load_scales(0, sa[0], sw[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kb = 0; kb < nk; kb += 2) {          // unrolled by two so the scale double buffer is register-indexed
    if (kb + 1 < nk) { load_tile(kb + 1, 1); load_scales(kb + 1, sa[1], sw[1]); }
    compute(0, sa[0], sw[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 1 >= nk) break;
    if (kb + 2 < nk) { load_tile(kb + 2, 0); load_scales(kb + 2, sa[0], sw[0]); }
    compute(1, sa[1], sw[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS fp32 [BM][BN] -> row vectors (gemm_epilogue.h)
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        const int col = wn * 32 * TN + j * 32 + frow;
        cs[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();

  const int n_out = p.act == TC_ACT_GEGLU ? p.n / 2 : p.n;
  if ((n_out & 7) != 0) epilogue_tail<BM, BN>(p, cs, tid, tile_m, tile_n, 0);
  else if (TN == 2 && p.act == TC_ACT_GEGLU) {
    if (p.alpha == 1.f && p.out_scale == 1.f) epilogue_fast<true, BM, BN, true>(p, cs, tid, tile_m, tile_n, 0);
    else epilogue_fast<true, BM, BN, false>(p, cs, tid, tile_m, tile_n, 0);
  }
  else if (p.alpha == 1.f && p.out_scale == 1.f && p.act == TC_ACT_NONE && !p.row_bias)
    epilogue_fast<false, BM, BN, true>(p, cs, tid, tile_m, tile_n, 0);
  else epilogue_fast<false, BM, BN, false>(p, cs, tid, tile_m, tile_n, 0);
}

// ---- bf16 -> MXFP8: one thread per 32-element block (64 input bytes, 32 output bytes, 1 scale byte)
__global__ __launch_bounds__(256) void quant_mx_kernel(const bf16_t* __restrict__ x, const int64_t rows, const int k, const int ld,
                                                       uint8_t* __restrict__ q, const int ldq, uint8_t* __restrict__ s, const int lds) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= rows * lds) return;
  const int64_t row = slot / lds;
  const int blk = (int)(slot - row * lds);
  if (blk * 32 >= k) { s[slot] = 0; return; }          // padding columns of the scale matrix
  const u32x4* src = reinterpret_cast<const u32x4*>(x + row * ld + blk * 32);
  u32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = src[i];
  uint32_t amax = 0;                                     // |x| as bf16 bits: integer order == magnitude order
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = v[i][e] & 0x7fffu, hi = (v[i][e] >> 16) & 0x7fffu;
      amax = max(amax, max(lo, hi));
    }
  // shared exponent = floor(log2(amax)) - 8 (e4m3 emax), clamped to E8M0's [-127, 127]; as a biased byte:
  const int e8 = (int)(amax >> 7);                       // bf16 exponent field of amax
  const int byte = e8 - 8 < 0 ? 0 : (e8 - 8 > 254 ? 254 : e8 - 8);
  const float inv = __uint_as_float((uint32_t)(254 - byte) << 23);   // 2^-(byte - 127), exact
  uint32_t out[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      float f[4];
      f[0] = __uint_as_float(v[i][e] << 16);
      f[1] = __uint_as_float(v[i][e] & 0xffff0000u);
      f[2] = __uint_as_float(v[i][e + 1] << 16);
      f[3] = __uint_as_float(v[i][e + 1] & 0xffff0000u);
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = fminf(fmaxf(f[t] * inv, -448.f), 448.f);   // saturate, then RNE to e4m3
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
      out[i * 2 + e / 2] = (uint32_t)w;
    }
  u32x4* dst = reinterpret_cast<u32x4*>(q + row * ldq + blk * 32);
  dst[0] = u32x4{out[0], out[1], out[2], out[3]};
  dst[1] = u32x4{out[4], out[5], out[6], out[7]};
  s[slot] = (uint8_t)byte;
}

}  // namespace

extern "C" int tc_quant_mxfp8(const tc_bf16* x, int64_t rows, int32_t k, int32_t ld, uint8_t* q, int32_t ldq,
                              uint8_t* s, int32_t lds, void* stream) {
  if (!x || !q || !s || rows <= 0 || k <= 0) return TC_EINVAL;
  if ((k & 31) || (ld & 7) || (ldq & 15) || !tc_aligned16(x) || !tc_aligned16(q)) return TC_EALIGN;
  if (ld < k || ldq < k || lds * 32 < k) return TC_ESHAPE;
  const int64_t slots = rows * lds;
  if ((slots + 255) / 256 > 0x7fffffffLL) return TC_ESHAPE;
  hipLaunchKernelGGL(quant_mx_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(x), rows, k, ld, q, ldq, s, lds);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_gemm_mxfp8(const TcGemmMxParams* pp, void* stream) {
  if (!pp) return TC_EINVAL;
  const TcGemmParams& p = pp->g;
  if (!p.a || !p.w || !p.c || !pp->a_scale || !pp->w_scale || p.m <= 0 || p.n <= 0 || p.k <= 0) return TC_EINVAL;
  if (!tc_aligned16(p.a) || !tc_aligned16(p.w) || !tc_aligned16(p.c)) return TC_EALIGN;
  if (p.residual && !tc_aligned16(p.residual)) return TC_EALIGN;
  if ((p.k & 31) || (p.lda & 15) || (p.ldw & 15)) return TC_EALIGN;
  if ((reinterpret_cast<uintptr_t>(pp->a_scale) & 1u) || (reinterpret_cast<uintptr_t>(pp->w_scale) & 3u)) return TC_EALIGN;
  if ((pp->lda_s & 1) || (pp->ldw_s & 3)) return TC_EALIGN;
  if (p.ldw < p.k || pp->ldw_s < ((p.k + 127) / 128) * 4) return TC_ESHAPE;
  if ((p.batch > 1) || p.stride_a || p.stride_w || p.stride_c) return TC_ESHAPE;
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  if ((n_out & 7) == 0) {
    if (p.out_f32 ? (p.ldc & 3) : (p.ldc & 7)) return TC_EALIGN;
    if (p.residual && (p.ldr & 7)) return TC_EALIGN;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15u)) return TC_EALIGN;
    if (p.row_bias && ((reinterpret_cast<uintptr_t>(p.row_bias) & 15u) || (p.ldrb & 3))) return TC_EALIGN;
  }
  if (p.ldc < n_out || (p.residual && p.ldr < n_out)) return TC_ESHAPE;
  if (geglu && ((p.n % 128) != 0 || p.row_bias || p.residual)) return TC_ESHAPE;
  if (p.row_bias && (p.row_div <= 0 || p.ldrb < p.n)) return TC_EINVAL;
  if (p.act < TC_ACT_NONE || p.act > TC_ACT_GEGLU) return TC_EINVAL;
  if (p.gather == TC_GATHER_LINEAR) {
    if (p.lda < p.k || pp->lda_s < ((p.k + 63) / 64) * 2) return TC_ESHAPE;
  } else if (p.gather == TC_GATHER_CONV3x3 || p.gather == TC_GATHER_CONVT3) {
    const int taps = p.gather == TC_GATHER_CONV3x3 ? 9 : 3;
    if (p.cin <= 0 || (p.cin % 64) != 0 || p.k != taps * p.cin || p.lda < p.cin || pp->lda_s * 32 < p.cin) return TC_ESHAPE;
    if ((int64_t)p.k * p.cin >= (1LL << 32)) return TC_ESHAPE;                      // tap decode by multiply-high
    if (p.frames <= 0 || p.h_out <= 0 || p.w_out <= 0) return TC_ESHAPE;
    if ((int64_t)p.frames * p.h_out * p.w_out != p.m) return TC_ESHAPE;
    if (p.gather == TC_GATHER_CONV3x3) {
      if (p.h_in <= 0 || p.w_in <= 0 || (p.stride != 1 && p.stride != 2)) return TC_ESHAPE;
      if (p.upsample && p.stride != 1) return TC_ESHAPE;
      const int hvv = p.upsample ? 2 * p.h_in : p.h_in, wvv = p.upsample ? 2 * p.w_in : p.w_in;
      if (p.pad != 0 && p.pad != 1) return TC_ESHAPE;
      const int extra = p.pad == 1 ? 2 : 1;
      if ((hvv + extra - 3) / p.stride + 1 != p.h_out || (wvv + extra - 3) / p.stride + 1 != p.w_out) return TC_ESHAPE;
    } else {
      if (p.t_len <= 0 || (p.frames % p.t_len) != 0) return TC_ESHAPE;
    }
  } else {
    return TC_EINVAL;
  }
  if (tc_a_rows(p) * p.lda >= 0x7fffff00LL || (int64_t)p.n * p.ldw >= 0x7fffff00LL) return TC_ESHAPE;   // 31-bit offsets
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // tile family as for the bf16 kernel: 64x64 where 128x128 tiles would leave most CUs idle
  const int64_t big_tiles = (int64_t)((p.n + 127) / 128) * ((p.m + 127) / 128);
  const bool small = !geglu && big_tiles < 384;
  const int bm = small ? 64 : 128;
  const int tiles_n = (p.n + bm - 1) / bm;
  const int tiles_m = (p.m + bm - 1) / bm;
  const int64_t nblk = (int64_t)tiles_n * 8 * ((tiles_m + 7) / 8);
  if (nblk > 0x7fffffffLL) return TC_ESHAPE;
  dim3 grid((unsigned)nblk), block(256);
  TcGemmParams half = p;                      // the walk heuristic prices W in bf16 bytes: fp8 rows are half as long
  half.ldw = p.ldw / 2;
  const int order = tc_gemm_tile_order(half, tiles_n);
#define TC_LAUNCH_MX(G)                                                                            \
  do {                                                                                             \
    if (small) hipLaunchKernelGGL((gemm_mx_kernel<G, 1, 1>), grid, block, 0, s, *pp, order);        \
    else hipLaunchKernelGGL((gemm_mx_kernel<G, 2, 2>), grid, block, 0, s, *pp, order);              \
  } while (0)
  switch (p.gather) {
    case TC_GATHER_LINEAR: TC_LAUNCH_MX(TC_GATHER_LINEAR); break;
    case TC_GATHER_CONV3x3: TC_LAUNCH_MX(TC_GATHER_CONV3x3); break;
    default: TC_LAUNCH_MX(TC_GATHER_CONVT3); break;
  }
#undef TC_LAUNCH_MX
  TC_LAUNCH_CHECK();
  return TC_OK;
}
