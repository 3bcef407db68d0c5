// 256x256-tile bf16 MFMA GEMM on FOUR waves -- one per SIMD, 128x128 outputs each, all 256 accumulators of a lane in AGPRs --
// for the linear problems whose tiles fill the chip: nn.Linear of lvdm/modules/attention.py:415-442 (the GEGLU projections and ff2
// of UNet levels 1 / 2), square problems.  gfx950.
//
// Why this shape.  Round 5 disassembled what hipBLASLt runs where it beats this library by 25-40 % (8192^3: 1.62 vs 1.23 PF/s;
// DESIGN.md 5.6 (4)): a 256x256x64 tile on four waves, each owning 8 x 8 v_mfma_f32_16x16x32_bf16 tiles.  Per K-step a wave
// issues 128 MFMAs from 32 ds_read_b128 (0.25 reads per MFMA: the 8-wave kernel of gemm8.hip needs 0.75, the 160-tile of
// gemm16.hip 0.4 -- and LDS bandwidth is the measured ceiling of both) and 16 LDS-DMA requests, and every one of those sits
// in the shadow of a 16-cycle MFMA, so ONE wave per SIMD keeps the matrix pipe busy: no second wave to share registers with,
// hence the 128x128 wave tile.  This kernel is that structure in HIP C++: gemm16.hip's two-fragment-set software pipeline
// (fragments of the next K-slice requested before the current slice's MFMAs; tile requests from inline asm between the MFMA
// rows; one counted wait + one raw barrier per K-step), scaled to 8 x 8 tiles per wave.
//
//   (the loop, its ring of five 32-KiB slice buffers and the waits are described at the loop)
// LDS: 5 slice buffers x (256 + 256) rows x 64 B = 160 KiB (one block per CU); XOR swizzle (row >> 1) & 3 on the SOURCE chunk of the
// DMA (64-byte rows: sixteen lanes of one chunk index hit eight 16-byte bank slots twice -- the minimum for a 256-byte read).
// Epilogue: per wave eight passes of one 16-row tile row through a private fp32 slab [16][128 + 4] in the idle stage buffers,
// then 16-byte row vectors (bias / row bias / activation / residual, or the GEGLU product of the per-32 packed columns).
// Linear gather only; the convolutions that dominate the UNet have N = 320 / 640 and would need a 256x160 variant with a
// gather -- not built.
#include "gemm_persist.h"

#include <stdlib.h>

namespace {

constexpr int G4_BM = 256, G4_BN = 256, G4_THREADS = 256;
constexpr int G4_WT = 128;                                   // wave tile
constexpr int G4_NT = G4_WT / 16;                            // 8 MFMA tiles per wave-tile side
constexpr int G4_KS = 32;                                    // K values per SLICE = one MFMA depth; the unit of the LDS ring
constexpr int G4_ROWB = G4_KS * 2;                           // bytes per tile row in a slice buffer: 64 (four 16-byte chunks)
constexpr int G4_HALF = G4_BM * G4_ROWB;                     // the A (or W) rows of one slice: 16 KiB
constexpr int G4_SB = 2 * G4_HALF;                           // one slice buffer: A rows | W rows = 32 KiB
constexpr int G4_NB = 5;                                     // ring of five slice buffers = 160 KiB: the whole CU
constexpr int G4_R = 4;                                      // request instructions per wave, slice and operand: 4 x 16 rows
constexpr int G4_SLAB_LD = G4_WT + 4;                        // fp32 slab row stride: +4 keeps the four row groups of a spill on distinct banks
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// The MFMA from inline asm with its accumulator pinned to AGPRs ("+a").  With the builtin, hipcc's register allocator split the
// 256 accumulators between AGPRs and VGPRs in the loop that also carries the request asm and moved them back and forth around
// every MFMA (392 v_accvgpr_* per 64 MFMAs in the ISA); pinned, a K-step is 128 MFMAs + 32 ds_read_b128 + 16 requests and
// nothing else.  Operand hazards the compiler cannot see through the asm: the operands come from ds_read_b128 (its waitcnt
// pass does guard asm register operands), the results are first read by VALU in the epilogue, behind a barrier.
__device__ __forceinline__ void g4_mfma(f32x4_t& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <bool GEGLU>
__global__ __launch_bounds__(G4_THREADS) void gemm4_kernel(const TcGemmParams p, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) char smem[G4_NB * G4_SB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int tiles_n = (p.n + G4_BN - 1) / G4_BN;
  const int tiles_m = (p.m + G4_BM - 1) / G4_BM;
  int tile_m, tile_n;
  g8_tile_of(blockIdx.x, tiles_m, tiles_n, total_tiles, tile_m, tile_n);

  const int64_t bz = blockIdx.z;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of smem
  const g8_srd_t w_srd = g8_make_srd(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));
  const g8_srd_t a_srd = g8_make_srd(reinterpret_cast<const bf16_t*>(p.a) + bz * p.stride_a + (int64_t)tile_m * G4_BM * p.lda,
                                     tc_a_extent(p) - (int64_t)tile_m * G4_BM * p.lda * 2);

  // ---- loader geometry.  One wave instruction fills 1 KiB = 16 rows x 64 B of a slice buffer, lane l at (row l >> 2, physical
  // chunk l & 3); the XOR swizzle (row >> 1) & 3 is applied to the SOURCE chunk.  Wave w requests rows (4 w + q) * 16 .. + 15 of
  // the A half and of the W half, q = 0..3: eight instructions per wave and slice.
  const int lr = lane >> 2;
  const int chunk = (lane & 3) ^ ((lr >> 1) & 3);              // the logical 16-byte chunk (8 K values) this lane fetches
  uint32_t a_voff[G4_R], b_voff[G4_R];
#pragma unroll
  for (int q = 0; q < G4_R; ++q) {
    const int ml = (wave * G4_R + q) * 16 + lr;
    a_voff[q] = tile_m * G4_BM + ml < p.m ? (uint32_t)((int64_t)ml * p.lda * 2 + chunk * 16) : TC_OOB;
    const int n = tile_n * G4_BN + ml;
    b_voff[q] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const int nh = p.k / G4_KS;                                  // slices of this problem (K % 64 == 0: host)

  uint32_t r_soff = 0, r_dst = 0;
  auto prep = [&](int h, int buf) {                            // buf = h % 5 (kept by the caller: no division in the loop)
    const int k0 = h * G4_KS;
    r_soff = (uint32_t)k0 * 2u;
    r_dst = lds0 + (uint32_t)(buf * G4_SB + wave_u * (G4_R * 1024));
  };
  // piece q of the prepared slice: 0..3 = W rows, 4..7 = A rows
  auto issue_piece = [&](auto Q_) {
    constexpr int q = decltype(Q_)::value;
    if constexpr (q < G4_R) g8_dma16(w_srd, r_dst + G4_HALF + q * 1024, b_voff[q], r_soff);
    else g8_dma16(a_srd, r_dst + (q - G4_R) * 1024, a_voff[q - G4_R], r_soff);
  };
  auto issue_all = [&]() {
    issue_piece(ic<0>{}); issue_piece(ic<1>{}); issue_piece(ic<2>{}); issue_piece(ic<3>{});
    issue_piece(ic<4>{}); issue_piece(ic<5>{}); issue_piece(ic<6>{}); issue_piece(ic<7>{});
  };

  f32x4_t acc[G4_NT][G4_NT];
#pragma unroll
  for (int i = 0; i < G4_NT; ++i)
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // v_mfma_f32_16x16x32_bf16 operands: lane holds row (lane & 15) of its 16-row tile, k = 8 (lane >> 4) .. +7 of the slice -> one
  // ds_read_b128 at chunk (lane >> 4) of that 64-byte row; tile rows step by 16, so (row >> 1) & 3 is the lane's.  Sixteen lanes
  // of one chunk index then touch eight distinct 16-byte bank slots twice: the two cycles a 256-byte read needs anyway.
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int f_col = (fq ^ ((frow >> 1) & 3)) << 4;
  const int a_row0 = (wm * G4_WT + frow) * G4_ROWB + f_col;
  const int b_row0 = G4_HALF + (wn * G4_WT + frow) * G4_ROWB + f_col;
  auto read_frags = [&](int buf, bf16x8 (&af)[G4_NT], bf16x8 (&bf)[G4_NT]) {
    const char* s0 = smem + buf * G4_SB;
#pragma unroll
    for (int i = 0; i < G4_NT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(s0 + a_row0 + i * (16 * G4_ROWB));
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(s0 + b_row0 + j * (16 * G4_ROWB));
  };
  // MFMA row i of a slice with the request (if any) of piece Q behind its eighth MFMA: six scalar / vector-memory instructions
  // issued in the shadow of the MFMA in flight.  MFMAs and requests are both volatile asm: this IS the instruction order.
  auto row = [&](auto I_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT], auto Q_) {
    constexpr int i = decltype(I_)::value, q = decltype(Q_)::value;
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    if constexpr (q >= 0) issue_piece(ic<(q >= 0 ? q : 0)>{});
  };

  // ---- main loop over SLICES (32 K values, 64 MFMAs per wave), a ring of five slice buffers.  At the top of slice h: its
  // fragments are in registers (set h & 1); slices h + 1 .. h + 4 are requested (buffers (h + 1 .. h + 4) % 5); buffer h % 5 has
  // been read by this wave.  Then
  //     s_waitcnt vmcnt(24): slice h + 1 has landed (the 3 x 8 newer requests stay in flight); s_barrier: for every wave, and
  //                          every wave is done with buffer h % 5
  //     slice h + 5 -> buffer h % 5 (8 requests, one behind each MFMA row) | fragments of slice h + 1 -> the other set |
  //     64 MFMAs of slice h
  // FOUR slices = two whole K-steps = 128 KiB per CU are in flight all the time (a two-stage scheme keeps 64 KiB: with an
  // L2 -> LDS round trip of ~2 us under load that, not the matrix pipe, set the pace of the first version of this kernel and of
  // gemm8.hip: 9 TB/s chip-wide where the library's own 256x256 kernel, with two K-steps in flight, draws 12.7).
  bf16x8 af0[G4_NT], bf0[G4_NT], af1[G4_NT], bf1[G4_NT];
  const int npre = nh < G4_NB ? nh : G4_NB;
  for (int h = 0; h < npre; ++h) {
    prep(h, h);
    issue_all();
  }
  // slice 0 has landed: everything requested after it may stay in flight
  if (npre >= 5) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if (npre == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (npre == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (npre == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  g8_barrier();
  read_frags(0, af0, bf0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // One slice.  STEADY: slice h + 5 exists and slices h + 1 .. h + 4 are all requested (compile-time waits, no branches);
  // otherwise the last slices of the tile: requests and waits by run-time conditions.
  auto slice = [&](auto STEADY_, int h, int buf, int nbuf, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT],
                   bf16x8 (&afn)[G4_NT], bf16x8 (&bfn)[G4_NT]) {
    constexpr bool STEADY = decltype(STEADY_)::value != 0;
    constexpr int N = -1;
    bool issue = true;
    if constexpr (STEADY) {
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else {
      issue = h + G4_NB < nh;
      const int newer = nh - 2 - h;                          // slices requested after h + 1 (3, 2, 1, 0; < 0: nothing to wait for)
      if (newer >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (newer == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    g8_barrier();
    if constexpr (STEADY) {
      prep(h + G4_NB, buf);
      row(ic<0>{}, af, bf, ic<0>{});
    } else {
      if (issue) {                                           // (all eight at once: the tail is five slices of a tile)
        prep(h + G4_NB, buf);
        issue_all();
      }
      row(ic<0>{}, af, bf, ic<N>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frags(nbuf, afn, bfn);                              // (behind the last slice it reads a dead buffer, nothing uses it)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STEADY) {
      row(ic<1>{}, af, bf, ic<1>{}); row(ic<2>{}, af, bf, ic<2>{}); row(ic<3>{}, af, bf, ic<3>{}); row(ic<4>{}, af, bf, ic<4>{});
      row(ic<5>{}, af, bf, ic<5>{}); row(ic<6>{}, af, bf, ic<6>{}); row(ic<7>{}, af, bf, ic<7>{});
    } else {
      row(ic<1>{}, af, bf, ic<N>{}); row(ic<2>{}, af, bf, ic<N>{}); row(ic<3>{}, af, bf, ic<N>{}); row(ic<4>{}, af, bf, ic<N>{});
      row(ic<5>{}, af, bf, ic<N>{}); row(ic<6>{}, af, bf, ic<N>{}); row(ic<7>{}, af, bf, ic<N>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments of slice h + 1 are in registers: this wave is done with buffer nbuf
  };
  // two slices per trip: the fragment sets alternate at compile time (K % 64 == 0: an even number of slices -- host)
  auto next = [&](int b) { return b == G4_NB - 1 ? 0 : b + 1; };
  int h = 0, buf = 0;
  for (; h + 1 + G4_NB < nh; h += 2) {
    const int b1 = next(buf), b2 = next(b1);
    slice(ic<1>{}, h, buf, b1, af0, bf0, af1, bf1);
    slice(ic<1>{}, h + 1, b1, b2, af1, bf1, af0, bf0);
    buf = b2;
  }
  for (; h < nh; h += 2) {
    const int b1 = next(buf), b2 = next(b1);
    slice(ic<0>{}, h, buf, b1, af0, bf0, af1, bf1);
    slice(ic<0>{}, h + 1, b1, b2, af1, bf1, af0, bf0);
    buf = b2;
  }
  __syncthreads();                                           // the epilogue slabs reuse the slice buffers

  // ---- epilogue: per wave, eight passes of one 16-row tile row through a private fp32 slab [16][132]
  float* slab = reinterpret_cast<float*>(smem) + wave * (16 * G4_SLAB_LD);
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * (p.out_f32 ? 4 : 2);
  const int n_out = GEGLU ? p.n / 2 : p.n;
  // this lane's 8 output columns are the same in every pass: vector column vc of the wave tile (GEGLU: of its 64 outputs)
  constexpr int VPR = GEGLU ? G4_WT / 16 : G4_WT / 8;        // output vectors per slab row: 8 | 16
  constexpr int QN = 16 * VPR / 64;                          // vectors per lane and pass: 2 | 4
  const int vc = lane % VPR, lr0 = lane / VPR;
  const int pc = GEGLU ? 32 * (vc >> 1) + (vc & 1) * 8 : vc * 8;          // slab column of the vector (GEGLU: of its VALUES; gates + 16)
  const int n0 = (GEGLU ? (tile_n * G4_BN + wn * G4_WT) / 2 : tile_n * G4_BN + wn * G4_WT) + vc * 8;
  const bool col_ok = n0 < n_out;
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_ok) {
    const float* bp = p.bias + (GEGLU ? tile_n * G4_BN + wn * G4_WT + pc : n0);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
    if (GEGLU) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(bp + 16), g1 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bg[e] = g0[e]; bg[4 + e] = g1[e]; }
    }
  }
  auto epi_pass = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < G4_NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(fq * 4 + r) * G4_SLAB_LD + j * 16 + frow] = acc[i][j][r];
    // the same wave reads back (LDS operations of one wave complete in order)
    const int row_base = tile_m * G4_BM + wm * G4_WT + i * 16;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int lr = lr0 + q * (64 / VPR);
      const int m = row_base + lr;
      if (m < p.m && col_ok) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * G4_SLAB_LD + pc);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * G4_SLAB_LD + pc + 4);
        float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (GEGLU) {
          const f32x4 glo = *reinterpret_cast<const f32x4*>(slab + lr * G4_SLAB_LD + pc + 16);
          const f32x4 ghi = *reinterpret_cast<const f32x4*>(slab + lr * G4_SLAB_LD + pc + 20);
          const float gt[8] = {glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
#pragma unroll
          for (int e = 0; e < 8; e += 2) {                   // the arithmetic of gemm_epilogue.h (packed fp32 pairs)
            const tc_f32x2 v = {x[e] * p.alpha + bv[e], x[e + 1] * p.alpha + bv[e + 1]};
            const tc_f32x2 h = v * gelu_erf_f2(tc_f32x2{gt[e] * p.alpha + bg[e], gt[e + 1] * p.alpha + bg[e + 1]}) * p.out_scale;
            x[e] = h[0]; x[e + 1] = h[1];
          }
        } else {
          float rb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (p.row_bias) {
            const float* rp = p.row_bias + (int64_t)(m / p.row_div) * p.ldrb + n0;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { rb[e] = r0[e]; rb[4 + e] = r1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = apply_act(x[e] * p.alpha + bv[e] + rb[e], p.act) * p.out_scale;
          if (res_base) {
            float rf[8];
            unpack8(*reinterpret_cast<const u32x4*>(res_base + (int64_t)m * p.ldr + n0), rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += rf[e];
          }
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
  };
  epi_pass(ic<0>{}); epi_pass(ic<1>{}); epi_pass(ic<2>{}); epi_pass(ic<3>{});
  epi_pass(ic<4>{}); epi_pass(ic<5>{}); epi_pass(ic<6>{}); epi_pass(ic<7>{});
}

int gemm4_mode() {         // TC_GEMM4 = 0 never | 1 / unset: the measured rule | 2 whenever the shape allows; read per call (A/B runs)
  const char* e = getenv("TC_GEMM4");
  return e ? atoi(e) : 1;
}

}  // namespace

// Decide whether the four-wave 256x256 kernel takes this (already validated) GEMM, and launch it.  1 = launched (or would be: dry).
int tc_gemm4_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry) {
  const int mode = gemm4_mode();
  if (mode == 0) return 0;
  if (p.gather != TC_GATHER_LINEAR || p.gn_part || p.a_norm || (p.n & 7) || p.k < 2 * TC_BK || (p.k % TC_BK) != 0 || p.k > 1024 * G4_KS) return 0;
  const bool geglu = p.act == TC_ACT_GEGLU;
  if (geglu && ((p.n & 31) || p.residual || p.row_bias)) return 0;
  if ((int64_t)G4_BM * p.lda * 2 >= 0x7fffff00LL || (int64_t)p.n * p.ldw * 2 >= 0x7fffff00LL) return 0;      // 31-bit offsets
  const int tiles_n = (p.n + G4_BN - 1) / G4_BN, tiles_m = (p.m + G4_BM - 1) / G4_BM;
  const int64_t total = (int64_t)tiles_n * tiles_m;
  if (total > 0x3fffffff || batch > 65535) return 0;
  if (mode == 1) {
    // the measured rule (profiles/r05_gemm4_bench.txt): TO BE FILLED IN from the first GPU measurement; until then nothing is routed
    return 0;
  }
  if (dry) return 1;
  dim3 grid((unsigned)total, 1, (unsigned)batch), block(G4_THREADS);
  if (geglu) hipLaunchKernelGGL(gemm4_kernel<true>, grid, block, 0, s, p, (int)total);
  else hipLaunchKernelGGL(gemm4_kernel<false>, grid, block, 0, s, p, (int)total);
  return 1;
}
