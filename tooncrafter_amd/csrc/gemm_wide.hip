// Wide-tile bf16 MFMA GEMM for the big-M layers, gfx950.
//
// Same contract as gemm.hip (implicit-GEMM gather, fused epilogue) with a 256-row block tile:
//   BM = 256, BN = 32*TNW*2 in {128, 256, 320}; 8 waves as 4(M) x 2(N), each wave owns a
//   64 x (32*TNW) patch = 2 x TNW v_mfma_f32_32x32x16_bf16 sub-tiles.
// Why: the 128x128 kernel moves 32 KiB into and 64 KiB out of LDS per K-step for 16 MFMAs per
// wave -- with two such blocks per CU the LDS pipe, not the MFMA pipe, sets the pace.  A 256x320
// tile writes 2.2x fewer LDS bytes per FLOP and reads 0.7 fragments per MFMA instead of 1.0, and
// BN = 320 divides every UNet width (320*k) exactly, so the N=320/960 layers of the highest
// resolution stop wasting 17 % of their MFMAs on padding columns.
// Pipeline: double-buffered LDS (2 x 72 KiB at BN=320) filled by direct global->LDS loads, the loads
// of K-step k+1 in flight under the 40 MFMAs per wave of K-step k, one barrier per K-step.  Epilogue: each wave
// transposes its accumulators through a private 10 KiB LDS slab, 16 rows at a time, and finishes
// on 16-byte row vectors (bias / row-bias / activation / GEGLU / residual / store).
#include "gemm_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int WBM = 256;
constexpr int WTHREADS = 512;

// PIPE: two K-steps of tile loads in flight, counted vmcnt + raw barriers (see gemm.hip)
template <int GATHER, int TNW, bool PIPE>
__global__ __launch_bounds__(WTHREADS, 2) void gemm_wide_kernel(const TcGemmParams p, const int order) {
  constexpr int BN = 64 * TNW;
  constexpr int WN = 32 * TNW;                         // columns per wave
  constexpr int STAGE_BYTES = (WBM + BN) * TC_BK * 2;  // 72 KiB at BN = 320
  constexpr int RA = WBM / 64, RB = BN / 64;           // loader rows per thread (64 rows per pass)
  constexpr int SLAB_FLOATS = 16 * WN;                 // per-wave epilogue slab: 16 rows x WN cols
  static_assert(8 * SLAB_FLOATS * 4 <= 2 * STAGE_BYTES, "epilogue slabs must fit in the pipeline buffers");
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform copy: LDS-DMA bases travel in M0

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tiles_m = (p.m + WBM - 1) / WBM;
  int tile_m, tile_n;
  tc_tile_of_block(blockIdx.x, tiles_m, tiles_n, order, tile_m, tile_n);
  if (tile_m >= tiles_m) return;

  const int64_t bz = blockIdx.z;
  const tc_rsrc_t w_rsrc = make_rsrc(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));

  // tile loads go global -> LDS directly (buffer_load_dwordx4 ... lds, see gemm.hip): lane l of a wave
  // instruction lands at byte 16 l of a 1-KiB piece = (row l>>3, physical chunk l&7) of 8 tile rows, so
  // the XOR swizzle is applied to the source chunk the lane fetches
  const int lrow = tid >> 3;     // 0..63
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  AGather<GATHER, RA> ag;
  ag.init(p, tile_m * WBM, lrow, 64, chunk);
  const tc_rsrc_t a_rsrc = tc_a_rsrc(p, bz, ag.row_lo);       // block-relative: 31-bit offsets span one tile's rows
  uint32_t b_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = tile_n * BN + lrow + 64 * i;
    b_voff[i] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const bool k_ragged = (p.k & (TC_BK - 1)) != 0;

  auto load_tile = [&](int kb, int stage) {
    const int k0 = kb * TC_BK;
    uint32_t a_voff[RA], a_soff;
    ag.offsets(p, k0, chunk, a_voff, a_soff);
    // OR-ing TC_OOB into an offset keeps it out of range: K-tail chunks are zero-filled without a branch
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    char* sa = smem + stage * STAGE_BYTES + wave_u * 1024;
    char* sb = sa + WBM * TC_BK * 2;
#pragma unroll
    for (int i = 0; i < RB; ++i) glds16(w_rsrc, sb + i * 8192, b_voff[i] | kill, (uint32_t)k0 * 2u);
#pragma unroll
    for (int i = 0; i < RA; ++i) glds16(a_rsrc, sa + i * 8192, a_voff[i] | kill, a_soff);
  };

  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  // fragment reads are double-buffered in registers where the budget allows (TNW <= 4): the ds_reads of
  // K-slice kk+1 are in flight under the MFMAs of slice kk; sched_barrier pins that order
  auto compute = [&](int stage) {
    const char* sa = smem + stage * STAGE_BYTES;
    const char* sb = sa + WBM * TC_BK * 2;
    auto frags = [&](int kk, bf16x8 (&a)[2], bf16x8 (&b)[TNW]) {
      const int c = kk * 2 + fhalf;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * 64 + i * 32 + frow, c));
#pragma unroll
      for (int j = 0; j < TNW; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(wn * WN + j * 32 + frow, c));
    };
    auto mfmas = [&](bf16x8 (&a)[2], bf16x8 (&b)[TNW]) {
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    if constexpr (TNW <= 4) {
      bf16x8 af[2][2], bf[2][TNW];
      frags(0, af[0], bf[0]);
      frags(1, af[1], bf[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(af[0], bf[0]);
      __builtin_amdgcn_sched_barrier(0);
      frags(2, af[0], bf[0]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(af[1], bf[1]);
      __builtin_amdgcn_sched_barrier(0);
      frags(3, af[1], bf[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(af[0], bf[0]);
      mfmas(af[1], bf[1]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 af[2], bf[TNW];
        frags(kk, af, bf);
        mfmas(af, bf);
      }
    }
  };

  // LDS-DMA data is visible to a ds_read only after the issuing wave's vmcnt wait AND a barrier the
  // reader has passed; the same barrier retires the reads of the stage the next iteration overwrites.
  const int nk = (p.k + TC_BK - 1) / TC_BK;
  if (PIPE) {
    load_tile(0, 0);
    if (nk > 1) load_tile(1, 1);
    for (int kb = 0; kb < nk; ++kb) {
      if (kb + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      compute(kb & 1);
      if (kb + 2 < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave has its fragments of this stage in registers
        load_tile(kb + 2, kb & 1);
      }
    }
    __syncthreads();                             // the epilogue slabs reuse the stage buffers
  } else {
    load_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
      if (kb + 1 < nk) load_tile(kb + 1, (kb + 1) & 1);
      compute(kb & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: per wave, 4 passes of 16 rows through a private fp32 slab
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  float* slab = reinterpret_cast<float*>(smem) + wave * SLAB_FLOATS;
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  const int col_w0 = tile_n * BN + wn * WN;          // first packed column of this wave

  // one pass = 16 rows; called with compile-time (i, half) so the accumulator indices stay static
  // (a runtime index would send the whole accumulator file to scratch)
  auto epi_pass = [&](auto I_, auto H_) {
    constexpr int i = decltype(I_)::value, half = decltype(H_)::value;
    // accumulator registers r = 8*half .. 8*half+7 hold local rows (r&3) + 4*fhalf + 8*((r>>2)&1)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 8 * half + q;
        const int lr = (r & 3) + 4 * fhalf + 8 * ((r >> 2) & 1);
        slab[lr * WN + j * 32 + frow] = acc[i][j][r];
      }
    // same wave reads back: LDS operations of one wave complete in order
    const int row_base = tile_m * WBM + wm * 64 + i * 32 + half * 16;
    if (!geglu) {
      constexpr int VPR = WN / 8;                    // 8-column vectors per slab row
      constexpr int NV = 16 * VPR / 64;              // vectors per lane (exact for WN in {64,128,160})
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int v = lane + 64 * q;
        const int lr = v / VPR, vc = v - lr * VPR;
        const int m = row_base + lr;
        const int n0 = col_w0 + vc * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * WN + vc * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * WN + vc * 8 + 4);
        if (m < p.m && n0 < p.n) {
          float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (p.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
          }
          if (p.row_bias) {
            const float* rp = p.row_bias + (int64_t)(m / p.row_div) * p.ldrb + n0;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] += r0[e]; bv[4 + e] += r1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = apply_act(x[e] * p.alpha + bv[e], p.act) * p.out_scale;
          if (res_base) {
            float rf[8];
            unpack8(*reinterpret_cast<const u32x4*>(res_base + (int64_t)m * p.ldr + n0), rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += rf[e];
          }
          if (p.out_f32) {
            float* op = reinterpret_cast<float*>(c_base) + (int64_t)m * p.ldc + n0;
            *reinterpret_cast<f32x4*>(op) = f32x4{x[0], x[1], x[2], x[3]};
            *reinterpret_cast<f32x4*>(op + 4) = f32x4{x[4], x[5], x[6], x[7]};
          } else {
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = pack8(x);
          }
        }
      }
    } else {
      // packed columns: every 32 = [16 values | 16 gates]; output vector u (8 columns) of a slab row
      // reads values at packed 32*(u/2) + 8*(u&1) and gates 16 further
      constexpr int VPR = WN / 16;                   // output vectors per slab row
      constexpr int TOT = 16 * VPR;
#pragma unroll
      for (int q = 0; q < (TOT + 63) / 64; ++q) {
        const int v = lane + 64 * q;
        const int lr = v / VPR, u = v - lr * VPR;
        const int m = row_base + lr;
        const int pc = 32 * (u >> 1) + 8 * (u & 1);            // packed column inside the wave's slab
        const int n0 = (col_w0 >> 1) + u * 8;                  // output column
        if (v < TOT && m < p.m && n0 < n_out) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * WN + pc);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * WN + pc + 4);
          const f32x4 glo = *reinterpret_cast<const f32x4*>(slab + lr * WN + pc + 16);
          const f32x4 ghi = *reinterpret_cast<const f32x4*>(slab + lr * WN + pc + 20);
          float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          float gt[8] = {glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
          float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (p.bias) {
            const float* bp = p.bias + col_w0 + pc;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(bp + 16), g1 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; bg[e] = g0[e]; bg[4 + e] = g1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; e += 2) {      // pairs: packed fp32 arithmetic (common.h gelu_erf_f2)
            const tc_f32x2 v = {x[e] * p.alpha + bv[e], x[e + 1] * p.alpha + bv[e + 1]};
            const tc_f32x2 h = v * gelu_erf_f2(tc_f32x2{gt[e] * p.alpha + bg[e], gt[e + 1] * p.alpha + bg[e + 1]}) * p.out_scale;
            x[e] = h[0]; x[e + 1] = h[1];
          }
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
  };
  using std::integral_constant;
  epi_pass(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  epi_pass(integral_constant<int, 0>{}, integral_constant<int, 1>{});
  epi_pass(integral_constant<int, 1>{}, integral_constant<int, 0>{});
  epi_pass(integral_constant<int, 1>{}, integral_constant<int, 1>{});
}

template <int TNW>
void launch_wide(const TcGemmParams& p, dim3 grid, hipStream_t s) {
  dim3 block(WTHREADS);
  const int order = tc_gemm_tile_order(p, (p.n + 64 * TNW - 1) / (64 * TNW));
  // TC_GEMM_PIPE = 2 only: with one 8-wave block per CU the second barrier per K-step costs more than the deeper
  // prefetch returns (measured, profiles/r03_pipe_bench.txt: the decoder's 512-channel convolutions 0.91x, the
  // N = 10240 GEGLU layer 0.95x), so the default keeps the plain loop
  const bool pipe = [] { const char* e = getenv("TC_GEMM_PIPE"); return e && e[0] == '2'; }();      // per call (A/B runs)
#define TC_LAUNCH_WIDE(G)                                                                           \
  do {                                                                                              \
    if (pipe) hipLaunchKernelGGL((gemm_wide_kernel<G, TNW, true>), grid, block, 0, s, p, order);    \
    else hipLaunchKernelGGL((gemm_wide_kernel<G, TNW, false>), grid, block, 0, s, p, order);        \
  } while (0)
  switch (p.gather) {
    case TC_GATHER_LINEAR: TC_LAUNCH_WIDE(TC_GATHER_LINEAR); break;
    case TC_GATHER_CONV3x3: TC_LAUNCH_WIDE(TC_GATHER_CONV3x3); break;
    default: TC_LAUNCH_WIDE(TC_GATHER_CONVT3); break;
  }
#undef TC_LAUNCH_WIDE
}

int wide_enabled() {        // TC_GEMM_WIDE=0: never (read per call: A/B runs flip it inside one process)
  const char* e = getenv("TC_GEMM_WIDE");
  return (e && e[0] == '0') ? 0 : 1;
}

}  // namespace

// Decide whether the wide kernel should take this (already validated) GEMM, and launch it.
int tc_gemm_wide_try(const TcGemmParams& p, int batch, hipStream_t s, bool force, bool dry) {
  if (!wide_enabled()) return 0;
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  if ((n_out & 7) != 0 || (p.n & 31) != 0) return 0;       // vector epilogue only; GEGLU packs per 32
  int tnw;
  if (p.n % 320 == 0) tnw = 5;
  else if (p.n >= 256) tnw = 4;
  else if (p.n >= 128) tnw = 2;
  else return 0;
  const int bn = 64 * tnw;
  const int tiles_n = (p.n + bn - 1) / bn;
  const int tiles_m = (p.m + WBM - 1) / WBM;
  const int64_t blocks = (int64_t)tiles_n * tiles_m * batch;
  // measured on MI355X (profiles/r01_v6_gemm_tile_sweep.txt): one 8-wave block per CU has no second block
  // to hide its prologue/epilogue behind, so the 256-row tile only wins with long K and either many
  // rounds of blocks (the decoder's 256/512-channel 3x3 convolutions) or a grid that fits the 256 CUs
  // once; wide GEGLU layers (N = 10240) win from K = 1280 because the 256x320 tile halves their
  // L2->LDS traffic
  if (!force) {
    const bool fits = blocks >= 192 && blocks <= 256;
    const bool conv_like = p.n >= 256 && p.k >= 2048 && (blocks >= 1024 || fits);
    // (r02: wide GEGLU layers -- N = 10240, K = 1280 -- took the 256x320 tile; with two K-steps in flight the 128x128 kernel
    // is ahead there too: 157-160 us vs 163-171 us, profiles/r03_pipe_bench.txt / r03_two_tiles_per_block_ab.txt)
    const bool wide_geglu = false;
    if (!conv_like && !wide_geglu) return 0;
  }
  const int64_t nblk = (int64_t)tiles_n * 8 * ((tiles_m + 7) / 8);
  if (nblk > 0x7fffffffLL) return 0;
  if (dry) return 1;
  dim3 grid((unsigned)nblk, 1, (unsigned)batch);
  if (tnw == 5) launch_wide<5>(p, grid, s);
  else if (tnw == 4) launch_wide<4>(p, grid, s);
  else launch_wide<2>(p, grid, s);
  return 1;
}
