// 160x160-tile bf16 MFMA GEMM for the UNet's level-0 / level-1 layers, gfx950.
//
// Same contract as gemm.hip (implicit-GEMM gather, fused epilogue, LDS-DMA tile loads, XOR-swizzled
// double-buffered LDS image, XCD-aware tile order); different TILE ARITHMETIC:
//   * every width of the ToonCrafter UNet is 320 * k and every row count 81920 / 4^level at B = 2, i.e. 5 * 2^n.
//     Power-of-two tiles never divide them: N = 320 runs three 128-column tiles (one of them half padding: -12 %)
//     and the 800 / 400 tiles of levels 1 / 2 fill the 512 resident block slots 1.56 / 0.78 times (-12...-19 %,
//     profiles/r02_quant_probe.txt).  A 160 x 160 tile divides them exactly: N = 320 -> 2 column tiles, level 0 ->
//     1024 tiles = 2.00 rounds of the 512 slots, level 1 (N = 640) -> 512 tiles = 1.00 round.
//   * 160 = 5 * 32 does not split over 2 x 2 waves with 32x32 MFMA tiles, so the wave tile is 80 x 80 =
//     5 x 5 v_mfma_f32_16x16x32_bf16 tiles (100 fp32 accumulators per lane): per 64-deep K-step a wave issues
//     50 MFMAs from 20 ds_read_b128 fragments (0.4 reads per MFMA; the 128x128 kernel needs 1.0), and the
//     block moves 40 KiB global -> LDS for 25600 MACs/row-of-K instead of 32 KiB for 16384 (-20 % L1 traffic
//     per FLOP).
//   * LDS: 2 stages x (160 + 160) rows x 128 B = 80 KiB: exactly two blocks per CU (160 KiB), as before.
// Epilogue: each wave transposes its accumulators through a private 5 KiB fp32 slab, one 16-row tile row at a
// time, and finishes on 16-byte row vectors (bias / row bias / activation / residual / store).  GEGLU layers
// (weights packed per 32 columns) stay on the 128x128 / 256x320 kernels.
#include "gemm_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int T16_BM = 160, T16_BN = 160;
constexpr int T16_STAGE = (T16_BM + T16_BN) * TC_BK * 2;     // 40 KiB
constexpr int T16_R = T16_BM / 32;                           // loader passes of 32 rows: 5 (A) + 5 (W)
constexpr int T16_WT = 80;                                   // wave tile
constexpr int T16_NT = T16_WT / 16;                          // 5 MFMA tiles per wave-tile side
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// PIPE: two K-steps of tile loads in flight, counted vmcnt + raw barriers (see gemm.hip)
// STATS (ABI 9, TcGemmParams.gn_part): the epilogue also emits, per output column, the sum and the sum of squares of
// the tile's 160 rows of bf16-rounded outputs.  Vector v = lane + 64 q of a 16-row pass is (row v / 10, column group
// v % 10) in EVERY pass, so a lane keeps one accumulator set per q (3 x 16 registers) and the walk of the plain
// epilogue stays as it is (a first version pinned lanes to column groups -- 60 active lanes, three row steps per pass --
// and cost the convolutions more than GroupNorm saved: +0.4 ms per forward against -0.3 ms).
template <int GATHER, bool PIPE, bool STATS>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(const TcGemmParams p, const int order) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * T16_STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int tiles_n = (p.n + T16_BN - 1) / T16_BN;
  const int tiles_m = (p.m + T16_BM - 1) / T16_BM;
  int tile_m, tile_n;
  tc_tile_of_block(blockIdx.x, tiles_m, tiles_n, order, tile_m, tile_n);
  if (tile_m >= tiles_m) return;

  const int64_t bz = blockIdx.z;
  const tc_rsrc_t w_rsrc = make_rsrc(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));

  // loader geometry as in gemm.hip: thread -> (row lrow + 32 i, 16-byte chunk); the swizzle (row>>1)&7 is
  // applied to the SOURCE chunk because the LDS destination of a DMA piece is lane-linear
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  AGather<GATHER, T16_R> ag;
  ag.init(p, tile_m * T16_BM, lrow, 32, chunk);
  const tc_rsrc_t a_rsrc = tc_a_rsrc(p, bz, ag.row_lo);       // block-relative: 31-bit offsets span one tile's rows
  uint32_t b_voff[T16_R];
#pragma unroll
  for (int i = 0; i < T16_R; ++i) {
    const int n = tile_n * T16_BN + lrow + 32 * i;
    b_voff[i] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const bool k_ragged = (p.k & (TC_BK - 1)) != 0;

  auto load_tile = [&](int kb, int stage) {
    const int k0 = kb * TC_BK;
    uint32_t a_voff[T16_R], a_soff;
    ag.offsets(p, k0, chunk, a_voff, a_soff);
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    char* sa = smem + stage * T16_STAGE + wave_u * 1024;
    char* sb = sa + T16_BM * TC_BK * 2;
#pragma unroll
    for (int i = 0; i < T16_R; ++i) glds16(w_rsrc, sb + i * 4096, b_voff[i] | kill, (uint32_t)k0 * 2u);
#pragma unroll
    for (int i = 0; i < T16_R; ++i) glds16(a_rsrc, sa + i * 4096, a_voff[i] | kill, a_soff);
  };

  f32x4_t acc[T16_NT][T16_NT];
#pragma unroll
  for (int i = 0; i < T16_NT; ++i)
#pragma unroll
    for (int j = 0; j < T16_NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // v_mfma_f32_16x16x32_bf16 operands: lane holds row (lane & 15) of its 16-row tile, k = 8 (lane >> 4) .. +7 of the
  // 32-deep slice -> one ds_read_b128 at 16-byte chunk 4 ks + (lane >> 4) of that row
  const int frow = lane & 15;
  const int fq = lane >> 4;
  int a_off[T16_NT], b_off[T16_NT];     // byte offsets of this lane's fragment rows for K-slice 0 (chunk fq)
#pragma unroll
  for (int i = 0; i < T16_NT; ++i) {
    a_off[i] = (wm * T16_WT + i * 16 + frow) * (TC_BK * 2);
    b_off[i] = (wn * T16_WT + i * 16 + frow) * (TC_BK * 2);
  }
  const int a_sw = ((wm * T16_WT + frow) >> 1) & 7;   // (row >> 1) & 7: tiles step by 16 rows, so i does not enter
  const int b_sw = ((wn * T16_WT + frow) >> 1) & 7;

  auto compute = [&](int stage) {
    const char* sa = smem + stage * T16_STAGE;
    const char* sb = sa + T16_BM * TC_BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[T16_NT], bf[T16_NT];
      const int ca = ((ks * 4 + fq) ^ a_sw) << 4;
      const int cb = ((ks * 4 + fq) ^ b_sw) << 4;
#pragma unroll
      for (int i = 0; i < T16_NT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + ca);
#pragma unroll
      for (int j = 0; j < T16_NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j] + cb);
#pragma unroll
      for (int i = 0; i < T16_NT; ++i)
#pragma unroll
        for (int j = 0; j < T16_NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = (p.k + TC_BK - 1) / TC_BK;
  if (PIPE) {
    load_tile(0, 0);
    if (nk > 1) load_tile(1, 1);
    for (int kb = 0; kb < nk; ++kb) {
      // stage kb & 1 has landed (the 2 * T16_R requests of the other stage may stay in flight), for every wave
      if (kb + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * T16_R) : "memory");
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

  // ---- epilogue: per wave, five passes of one 16-row tile row through a private fp32 slab [16][80]
  float* slab = reinterpret_cast<float*>(smem) + wave * (16 * T16_WT);
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  const int col_w0 = tile_n * T16_BN + wn * T16_WT;
  constexpr int VPR = T16_WT / 8;                      // 10 vectors of 8 columns per slab row
  float gs[3][8], gq[3][8];                            // STATS: column sums of vector slot lane + 64 q over the five passes
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[q][e] = 0.f; gq[q][e] = 0.f; }
  auto epi_pass = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < T16_NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(fq * 4 + r) * T16_WT + j * 16 + frow] = acc[i][j][r];
    // the same wave reads back (LDS operations of one wave complete in order): 160 vectors over 64 lanes
    const int row_base = tile_m * T16_BM + wm * T16_WT + i * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int v = lane + 64 * q;
      const int lr = v / VPR, vc = v - lr * VPR;
      const int m = row_base + lr;
      const int n0 = col_w0 + vc * 8;
      if (v < 16 * VPR && m < p.m && n0 < p.n) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * T16_WT + vc * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * T16_WT + vc * 8 + 4);
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
          const u32x4 packed = pack8(x);
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = packed;
          if (STATS) {
            float fr[8];
            unpack8(packed, fr);                     // what GroupNorm will read back: the rounded values
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[q][e] += fr[e]; gq[q][e] += fr[e] * fr[e]; }
          }
        }
      }
    }
  };
  using std::integral_constant;
  epi_pass(integral_constant<int, 0>{});
  epi_pass(integral_constant<int, 1>{});
  epi_pass(integral_constant<int, 2>{});
  epi_pass(integral_constant<int, 3>{});
  epi_pass(integral_constant<int, 4>{});
  if (STATS) {
    // fold: 16 vector slots (rows of a pass) x 2 waves (wm) per column, through LDS in a fixed order (no atomics).
    // Each wave parks its 160 slots x 16 floats in a 10 KiB region of the (now idle) stage buffers.
    __syncthreads();                                   // every wave is done with its epilogue slab
    float* mine = reinterpret_cast<float*>(smem) + wave * (160 * 16);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int v = lane + 64 * q;
      if (v < 16 * VPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mine[v * 16 + e] = gs[q][e]; mine[v * 16 + 8 + e] = gq[q][e]; }
      }
    }
    __syncthreads();
    if (tid < T16_BN) {
      const int n = tile_n * T16_BN + tid;
      if (n < p.n) {
        const int wnn = tid / T16_WT, cw = tid - wnn * T16_WT;       // wave column, column inside the wave tile
        const int vc = cw >> 3, e = cw & 7;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int wmm = 0; wmm < 2; ++wmm) {
          const float* sl = reinterpret_cast<const float*>(smem) + (wmm * 2 + wnn) * (160 * 16);
#pragma unroll
          for (int lr = 0; lr < 16; ++lr) { a += sl[(lr * VPR + vc) * 16 + e]; b += sl[(lr * VPR + vc) * 16 + 8 + e]; }
        }
        float* o = p.gn_part + (int64_t)tile_m * 2 * p.n + n;
        o[0] = a;
        o[p.n] = b;
      }
    }
  }
}

int tile16_mode() {        // TC_GEMM_TILE16 = 0 never | 1 heuristic (default) | 2 whenever the shape allows
  const char* e = getenv("TC_GEMM_TILE16");     // read per call: the parity tests flip it inside one process
  return e ? atoi(e) : 1;
}

}  // namespace

// Decide whether the 160x160 kernel should take this (already validated) GEMM, and launch it.  1 = launched.
int tc_gemm_tile16_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry) {
  const int mode = tile16_mode();
  if (mode == 0 || p.act == TC_ACT_GEGLU || (p.n % T16_BN) != 0) return 0;
  const int tiles_n = p.n / T16_BN;
  const int tiles_m = (p.m + T16_BM - 1) / T16_BM;
  const int64_t tiles = (int64_t)tiles_n * tiles_m * batch;
  if (mode == 1) {
    // worth it where the 160-grid quantises better than the 128-grid onto the 512 resident block slots:
    // efficiency = useful work / (whole rounds of 512 blocks x padded tile area)
    const int64_t t128 = (int64_t)((p.n + 127) / 128) * ((p.m + 127) / 128) * batch;
    const double eff16 = (double)tiles / (double)((tiles + 511) / 512 * 512) * ((double)p.m / ((double)tiles_m * T16_BM));
    const double eff128 = (double)t128 / (double)((t128 + 511) / 512 * 512) *
                          ((double)p.n / (double)((p.n + 127) / 128 * 128)) *
                          ((double)p.m / (double)((p.m + 127) / 128 * 128));
    // measured (profiles/r02_tile16_ab.txt, r02_tile16_cold_sustained.txt): 3x3 / temporal convolutions gain
    // 1.12-1.32x (cache-hot, cache-cold and sustained alike), K >= 1280 linear layers 1.0-1.18x, short-K projections
    // (K <= 640: five-pass epilogue per 160x160 tile against 5-10 K-steps) LOSE 8-18 %.  The heuristic takes the
    // convolutions only: the linear layers' gain is within the run-to-run spread of a whole clip.
    if (tiles < 512 || eff16 < eff128 + 0.04 || p.k < 960 || p.gather == TC_GATHER_LINEAR) return 0;
  }
  const int64_t nblk = (int64_t)tiles_n * 8 * ((tiles_m + 7) / 8);
  if (nblk > 0x7fffffffLL) return 0;
  if (dry) return 1;
  dim3 grid((unsigned)nblk, 1, (unsigned)batch), block(256);
  const int order = tc_gemm_tile_order(p, tiles_n);
  const bool stats = p.gn_part != nullptr;
  // TC_GEMM_PIPE = 2 only: on this tile the deeper prefetch measured 0.97-1.02x (profiles/r03_pipe_bench.txt) -- two
  // blocks of 80 KiB per CU already overlap each other's load latency -- so the default keeps the plain loop
  const bool pipe = [] { const char* e = getenv("TC_GEMM_PIPE"); return e && e[0] == '2'; }();      // per call (A/B runs)
#define TC_LAUNCH16(G)                                                                              \
  do {                                                                                              \
    if (stats) hipLaunchKernelGGL((gemm16_kernel<G, false, true>), grid, block, 0, s, p, order);    \
    else if (pipe) hipLaunchKernelGGL((gemm16_kernel<G, true, false>), grid, block, 0, s, p, order); \
    else hipLaunchKernelGGL((gemm16_kernel<G, false, false>), grid, block, 0, s, p, order);         \
  } while (0)
  switch (p.gather) {
    case TC_GATHER_LINEAR: TC_LAUNCH16(TC_GATHER_LINEAR); break;
    case TC_GATHER_CONV3x3: TC_LAUNCH16(TC_GATHER_CONV3x3); break;
    default: TC_LAUNCH16(TC_GATHER_CONVT3); break;
  }
#undef TC_LAUNCH16
  return 1;
}
