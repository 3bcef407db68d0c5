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
#include "gemm_persist.h"

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
// ABL (TC_G16_ABLATE, timing builds of the 3x3 gather only -- the results are WRONG): 1 = the A tile is requested for
// the dx = -1 tap of each kernel row only (the L2->LDS traffic a slab shared by the three dx taps would have), 2 = no A
// requests, 3 = no W requests, 4 = no requests at all (MFMAs + fragment reads + epilogue)
// ILV (TC_G16_ILV): the tile requests of the K loop issued ONE OR TWO AT A TIME between the MFMAs instead of as a burst
// of ten per wave in front of them.  Why: a wave cannot issue MFMAs while its LDS-DMA instructions queue for the CU's
// address unit (~23 clk per 1 KiB piece with all eight waves streaming, scripts/probes/load_probe.hip), so the burst is a
// phase of ~900 clk per block and K-step in which the block's four waves compute nothing -- loads-only 0.77 us and
// MFMAs-only 0.99 us per K-step pair ADD to the measured 1.63 us instead of overlapping (profiles/r04_g16_ablate.txt).
//   1 = the plain loop (request step k + 1, compute step k, wait, barrier) with the requests spread over the first half
//       of the step's MFMAs;
//   2 = a software pipeline over two LDS stages with TWO fragment sets: per K-step [fragments of the second K-slice |
//       25 MFMAs of the first | wait: own reads done + own pieces of step k + 1 landed | barrier | fragments of step
//       k + 1's first K-slice | 25 MFMAs of the second with the ten requests of step k + 2 between them] -- a request
//       has half a step to land, a fragment read a quarter.
// Requests come from inline asm (gemm_persist.h: g8_dma16): hipcc neither reorders them nor guards fragment reads of
// the other stage with vmcnt(0).  Same MFMA order per accumulator as the plain loop: bit-identical results.
// WM = 4 (TC_G16_TALL): a 320 x 160 tile on EIGHT waves (4 x 2 of the same 80 x 80 wave tiles), one block per CU.  The W
// tile is what a K-step pays for (without its requests the kernel runs at its MFMA-only time, without A's it does not:
// profiles/r04_g16_ablate.txt); a tall block requests it once for twice the rows -- 60 KiB per K-step and CU instead of
// 80, W's share halved -- and keeps the wave tile, the fragment reads per MFMA and the epilogue as they are.
template <int GATHER, bool PIPE, bool STATS, int ABL = 0, int ILV = 0, int WM = 2>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void gemm16_kernel(const TcGemmParams p, const int order) {
  constexpr int BM = T16_WT * WM;                          // tile rows: 160 | 320
  constexpr int RSTEP = 16 * WM;                           // rows per loader pass (threads / 8): 32 | 64
  constexpr int RA = BM / RSTEP;                           // loader passes over the A rows: 5
  constexpr int RB = (T16_BN + RSTEP - 1) / RSTEP;         // ... over the W rows: 5 | 3 (the third half empty)
  constexpr int STAGE = (BM + T16_BN) * TC_BK * 2;         // 40 | 60 KiB
  constexpr int PIECE = RSTEP * TC_BK * 2;                 // LDS bytes from one loader pass to the next
  static_assert(WM == 2 || (!STATS && ABL == 0), "statistics and timing builds exist for the 160-row tile only");
  // WM = 4: the third W pass covers rows 128..191 of a 160-row tile.  Its upper half (waves 4..7) requests nothing real
  // (offsets out of range -> zeros), but a request it must be -- the counted waits assume the same number per wave -- and
  // zeros written behind the W rows would land in the next stage's A rows: those four pieces go to a 4 KiB dump
  constexpr int DUMP = WM == 4 ? 4096 : 0;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE + DUMP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int tiles_n = (p.n + T16_BN - 1) / T16_BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  int tile_m, tile_n;
  tc_tile_of_block(blockIdx.x, tiles_m, tiles_n, order, tile_m, tile_n);
  if (tile_m >= tiles_m) return;

  const int64_t bz = blockIdx.z;
  const tc_rsrc_t w_rsrc = make_rsrc(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));

  // loader geometry as in gemm.hip: thread -> (row lrow + 32 i, 16-byte chunk); the swizzle (row>>1)&7 is
  // applied to the SOURCE chunk because the LDS destination of a DMA piece is lane-linear
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  AGather<GATHER, RA> ag;
  ag.init(p, tile_m * BM, lrow, RSTEP, chunk);
  const tc_rsrc_t a_rsrc = tc_a_rsrc(p, bz, ag.row_lo);       // block-relative: 31-bit offsets span one tile's rows
  uint32_t b_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int nl = lrow + RSTEP * i;                        // (the tall tile's third pass covers W rows 128..191: 160.. are no rows)
    const int n = tile_n * T16_BN + nl;
    b_voff[i] = (nl < T16_BN && n < p.n) ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const bool k_ragged = (p.k & (TC_BK - 1)) != 0;

  auto load_tile = [&](int kb, int stage) {
    const int k0 = kb * TC_BK;
    uint32_t a_voff[RA], a_soff;
    ag.offsets(p, k0, chunk, a_voff, a_soff);
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    char* sa = smem + stage * STAGE + wave_u * 1024;
    char* sb = sa + BM * TC_BK * 2;
    if (ABL != 3 && ABL != 4) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        char* d = sb + i * PIECE;
        if (WM == 4 && i == RB - 1 && wave_u >= 4) d = smem + 2 * STAGE + (wave_u - 4) * 1024;
        glds16(w_rsrc, d, b_voff[i] | kill, (uint32_t)k0 * 2u);
      }
    }
    if (ABL == 0 || ABL == 3 || (ABL == 1 && ((k0 / p.cin) % 3) == 0)) {
#pragma unroll
      for (int i = 0; i < RA; ++i) glds16(a_rsrc, sa + i * PIECE, a_voff[i] | kill, a_soff);
    }
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
    const char* sa = smem + stage * STAGE;
    const char* sb = sa + BM * TC_BK * 2;
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
  if constexpr (ILV != 0) {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of smem
    const g8_srd_t w_srd = g8_make_srd(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));
    // the slim request state of the persistent kernels (per row a byte offset and a tap mask; stride-1 3x3 only: the
    // host keeps stride 2 / fused upsample on the plain loop) -- AGather's generic path costs 15 more registers
    G8Gather<GATHER, RSTEP, RA> sg;
    sg.init(p, tile_m * BM, lrow, chunk);
    const g8_srd_t a_srd = g8_make_srd(reinterpret_cast<const bf16_t*>(p.a) + bz * p.stride_a + sg.row_lo * p.lda,
                                       tc_a_extent(p) - sg.row_lo * p.lda * 2);
    const int tpt = GATHER == TC_GATHER_LINEAR ? 1 : p.cin / TC_BK;     // K-tiles per tap (cin % 64 == 0: host)
    const uint32_t tap_magic = (65536u + tpt - 1) / tpt;                // kb / tpt as a multiply-high, exact for kb < 1024
    // the RB + RA (ten | eight) requests of one K-step: pieces 0..RB-1 = W row passes, then the A row passes
    uint32_t r_avoff[RA], r_asoff = 0, r_wsoff = 0, r_dst = 0;
    int r_tap = -1;                                       // the tap r_avoff was formed for: the row offsets change with the TAP
    auto prep = [&](int kb, int stage) {                  // (every cin / 64 K-steps), only the scalar offset with the K-step
      const int k0 = kb * TC_BK;
      int tap = 0;
      r_asoff = (uint32_t)k0 * 2u;
      if (GATHER != TC_GATHER_LINEAR) {
        tap = (int)(((uint32_t)kb * tap_magic) >> 16);
        r_asoff = (uint32_t)(k0 - tap * p.cin) * 2u;
      }
      uint32_t kill = 0u;
      if (k_ragged && kb == nk - 1) {                     // the K tail: its chunks are zero-filled by an out-of-range offset.
        kill = (k0 + chunk * 8 >= p.k) ? TC_OOB : 0u;     // Only the tile's LAST requests see it, so W's offsets take it for good
#pragma unroll
        for (int i = 0; i < RB; ++i) b_voff[i] |= kill;
        r_tap = -1;
      }
      if (tap != r_tap) {
        r_tap = tap;
        uint32_t delta = 0;
        if (GATHER == TC_GATHER_CONV3x3) {
          const int ty = (tap * 11) >> 5;                               // tap / 3 for tap < 9
          delta = (uint32_t)(((ty - 1) * p.w_in + (tap - ty * 3 - 1)) * p.lda * 2);
        } else if (GATHER == TC_GATHER_CONVT3) {
          delta = (uint32_t)((tap - 1) * p.h_out * p.w_out * p.lda * 2);
        }
#pragma unroll
        for (int q = 0; q < RA; ++q) r_avoff[q] = sg.voff(q, tap, delta) | kill;
      }
      r_wsoff = (uint32_t)k0 * 2u;
      r_dst = lds0 + (uint32_t)(stage * STAGE + wave_u * 1024);
    };
    // piece q of the prepared K-step: (descriptor, LDS destination, row offset, scalar offset)
    auto q_dst = [&](auto Q_) -> uint32_t {
      constexpr int q = decltype(Q_)::value;
      if constexpr (q < RB) {
        uint32_t d = r_dst + BM * TC_BK * 2 + q * PIECE;
        if (WM == 4 && q == RB - 1 && wave_u >= 4) d = lds0 + 2 * STAGE + (wave_u - 4) * 1024;      // (see DUMP)
        return d;
      } else {
        return r_dst + (q - RB) * PIECE;
      }
    };
    auto issue_pair = [&](auto I_) {                    // the two pieces that go behind MFMA row i (the last pair may be short)
      constexpr int q0 = 2 * decltype(I_)::value, q1 = q0 + 1;
      if constexpr (q1 < RB) {
        g8_dma16x2(w_srd, q_dst(ic<q0>{}), b_voff[q0], r_wsoff, w_srd, q_dst(ic<q1>{}), b_voff[q1], r_wsoff);
      } else if constexpr (q0 < RB) {
        g8_dma16x2(w_srd, q_dst(ic<q0>{}), b_voff[q0], r_wsoff, a_srd, q_dst(ic<q1>{}), r_avoff[q1 - RB], r_asoff);
      } else if constexpr (q1 < RB + RA) {
        g8_dma16x2(a_srd, q_dst(ic<q0>{}), r_avoff[q0 - RB], r_asoff, a_srd, q_dst(ic<q1>{}), r_avoff[q1 - RB], r_asoff);
      } else if constexpr (q0 < RB + RA) {
        g8_dma16(a_srd, q_dst(ic<q0>{}), r_avoff[q0 - RB], r_asoff);
      }
    };
    auto issue_all = [&]() {
      issue_pair(ic<0>{}); issue_pair(ic<1>{}); issue_pair(ic<2>{}); issue_pair(ic<3>{}); issue_pair(ic<4>{});
    };
    auto read_frags = [&](int stage, int ks, bf16x8 (&af)[T16_NT], bf16x8 (&bf)[T16_NT]) {
      const char* sa = smem + stage * STAGE;
      const char* sb = sa + BM * TC_BK * 2;
      const int ca = ((ks * 4 + fq) ^ a_sw) << 4;
      const int cb = ((ks * 4 + fq) ^ b_sw) << 4;
#pragma unroll
      for (int i = 0; i < T16_NT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + ca);
#pragma unroll
      for (int j = 0; j < T16_NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j] + cb);
    };
    auto mma_row = [&](auto I_, const bf16x8 (&af)[T16_NT], const bf16x8 (&bf)[T16_NT]) {
      constexpr int i = decltype(I_)::value;
#pragma unroll
      for (int j = 0; j < T16_NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    prep(0, 0);
    issue_all();
    if constexpr (ILV == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g8_barrier();
      for (int kb = 0; kb < nk; ++kb) {
        const int st = kb & 1;
        const bool more = kb + 1 < nk;
        if (more) prep(kb + 1, st ^ 1);
        bf16x8 af[T16_NT], bf[T16_NT];
        read_frags(st, 0, af, bf);
        auto row0 = [&](auto I_) {
          mma_row(I_, af, bf);
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_pair(I_);
          __builtin_amdgcn_sched_barrier(0);
        };
        row0(ic<0>{}); row0(ic<1>{}); row0(ic<2>{}); row0(ic<3>{}); row0(ic<4>{});
        read_frags(st, 1, af, bf);
        mma_row(ic<0>{}, af, bf); mma_row(ic<1>{}, af, bf); mma_row(ic<2>{}, af, bf); mma_row(ic<3>{}, af, bf); mma_row(ic<4>{}, af, bf);
        __builtin_amdgcn_sched_barrier(0);           // (hipcc otherwise sinks the MFMAs below the wait: it is no memory operation to them)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        g8_barrier();
      }
    } else {
      if (nk > 1) {
        prep(1, 1);
        issue_all();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");         // step 0 has landed, step 1 may be in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      g8_barrier();
      bf16x8 af0[T16_NT], bf0[T16_NT], af1[T16_NT], bf1[T16_NT];
      read_frags(0, 0, af0, bf0);
      for (int kb = 0; kb < nk; ++kb) {
        const int st = kb & 1;
        const bool more2 = kb + 2 < nk;
        // first K-slice: its fragments were read during the previous step; the second slice's are read beside its MFMAs
        read_frags(st, 1, af1, bf1);
        mma_row(ic<0>{}, af0, bf0); mma_row(ic<1>{}, af0, bf0); mma_row(ic<2>{}, af0, bf0); mma_row(ic<3>{}, af0, bf0); mma_row(ic<4>{}, af0, bf0);
        // own reads of stage st complete; own pieces of step kb + 1 (requested half a step ago) landed
        __builtin_amdgcn_sched_barrier(0);           // (hipcc otherwise sinks the MFMAs below the wait: it is no memory operation to them)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        g8_barrier();                                  // -> stage st is free, stage st ^ 1 is complete, for every wave
        if (more2) prep(kb + 2, st);
        read_frags(st ^ 1, 0, af0, bf0);              // (unconditional: in the last step it reads a dead stage and nothing uses it;
                                                      //  under a branch hipcc's waitcnt pass makes the MFMAs below wait for these reads)
        auto row1 = [&](auto I_) {
          mma_row(I_, af1, bf1);
          __builtin_amdgcn_sched_barrier(0);
          if (more2) issue_pair(I_);
          __builtin_amdgcn_sched_barrier(0);
        };
        row1(ic<0>{}); row1(ic<1>{}); row1(ic<2>{}); row1(ic<3>{}); row1(ic<4>{});
      }
    }
    __syncthreads();                               // the epilogue slabs reuse the stage buffers
  } else if (PIPE) {
    load_tile(0, 0);
    if (nk > 1) load_tile(1, 1);
    for (int kb = 0; kb < nk; ++kb) {
      // stage kb & 1 has landed (the RA + RB requests of the other stage may stay in flight), for every wave
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
    const int row_base = tile_m * BM + wm * T16_WT + i * 16;
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
#ifdef TC_TIMING_BUILDS      /* timing ablations: WRONG results by construction, never in the product library */
  const int abl = [] { const char* e = getenv("TC_G16_ABLATE"); return e ? atoi(e) : 0; }();             // per call (timing runs)
  if (abl >= 1 && abl <= 4 && p.gather == TC_GATHER_CONV3x3 && !stats) {
    if (abl == 1) hipLaunchKernelGGL((gemm16_kernel<TC_GATHER_CONV3x3, false, false, 1>), grid, block, 0, s, p, order);
    if (abl == 2) hipLaunchKernelGGL((gemm16_kernel<TC_GATHER_CONV3x3, false, false, 2>), grid, block, 0, s, p, order);
    if (abl == 3) hipLaunchKernelGGL((gemm16_kernel<TC_GATHER_CONV3x3, false, false, 3>), grid, block, 0, s, p, order);
    if (abl == 4) hipLaunchKernelGGL((gemm16_kernel<TC_GATHER_CONV3x3, false, false, 4>), grid, block, 0, s, p, order);
    return 1;
  }
#endif
  // TC_G16_ILV: requests between the MFMAs (see the kernel header).  Unset = loop 2 for the convolutions, the plain loop
  // for linear problems (measured, profiles/r04_g16_tall_ilv_bench.txt: 3x3 / temporal convolutions 1.01-1.03x on the
  // 160-row tile, linear 0.98-1.02x; inside the UNet 8.29 / 8.30 against 8.27 / 8.28 frames/s, alternating on one lease:
  // profiles/r04_clip_ab_g16.txt); 0 = never; 1 | 2 = that loop wherever it can run.  Read per call (A/B runs).
  const bool ilv_ok = (p.gather == TC_GATHER_LINEAR || ((p.cin % TC_BK) == 0 && p.cin <= 4096)) && p.k < 1024 * TC_BK &&
                      (p.gather != TC_GATHER_CONV3x3 || (p.stride == 1 && !p.upsample && p.pad == 1));
  const int ilv = [&] { const char* e = getenv("TC_G16_ILV"); return e ? atoi(e) : (p.gather != TC_GATHER_LINEAR ? 2 : 0); }();
  // TC_G16_TALL: the 320 x 160 tile on eight waves (WM = 4), one block per CU.  0 (default) = never; 1 = convolutions whose
  // tall tiles fill whole rounds of the 256 CUs as well as the 160-row tiles fill their 512 slots (levels 0 / 1 of the
  // UNet); 2 = whenever the shape allows.  OFF although it wins kernel by kernel (with loop 2: level 0 1.035-1.06x, level
  // 1 1.09x; level 2 has 128 tall tiles: 0.8x; linear problems 0.94-1.05x): inside the UNet, alternating on one lease, it
  // LOSES 0.8 % of a clip (8.22 vs 8.27-8.29 frames/s, and 7.32 vs 7.38 together with loop 2:
  // profiles/r04_clip_ab_g16.txt) -- one 8-wave block per CU has nothing to fill its ramp and tail with.
  const int tall = [] { const char* e = getenv("TC_G16_TALL"); return e ? atoi(e) : 0; }();
  if (tall >= 1 && !stats && (ilv == 0 || ilv_ok) && (tall == 2 || p.gather != TC_GATHER_LINEAR)) {
    const int tm4 = (p.m + 2 * T16_BM - 1) / (2 * T16_BM);
    const int64_t t4 = (int64_t)tiles_n * tm4 * batch;
    const double e4 = (double)t4 / (double)((t4 + 255) / 256 * 256) * ((double)p.m / ((double)tm4 * 2 * T16_BM));
    const double e2 = (double)tiles / (double)((tiles + 511) / 512 * 512) * ((double)p.m / ((double)tiles_m * T16_BM));
    if (tall == 2 || (t4 >= 256 && e4 >= e2 - 0.02)) {
      const int64_t nb4 = (int64_t)tiles_n * 8 * ((tm4 + 7) / 8);
      dim3 grid4((unsigned)nb4, 1, (unsigned)batch), block4(512);
#define TC_LAUNCH16_TALL(G)                                                                                          \
  do {                                                                                                               \
    if (ilv == 1) hipLaunchKernelGGL((gemm16_kernel<G, false, false, 0, 1, 4>), grid4, block4, 0, s, p, order);      \
    else if (ilv == 2) hipLaunchKernelGGL((gemm16_kernel<G, false, false, 0, 2, 4>), grid4, block4, 0, s, p, order); \
    else hipLaunchKernelGGL((gemm16_kernel<G, false, false, 0, 0, 4>), grid4, block4, 0, s, p, order);               \
  } while (0)
      switch (p.gather) {
        case TC_GATHER_LINEAR: TC_LAUNCH16_TALL(TC_GATHER_LINEAR); break;
        case TC_GATHER_CONV3x3: TC_LAUNCH16_TALL(TC_GATHER_CONV3x3); break;
        default: TC_LAUNCH16_TALL(TC_GATHER_CONVT3); break;
      }
#undef TC_LAUNCH16_TALL
      return 1;
    }
  }
  if ((ilv == 1 || ilv == 2) && !stats && ilv_ok) {
#define TC_LAUNCH16_ILV(G)                                                                                       \
  do {                                                                                                           \
    if (ilv == 1) hipLaunchKernelGGL((gemm16_kernel<G, false, false, 0, 1>), grid, block, 0, s, p, order);       \
    else hipLaunchKernelGGL((gemm16_kernel<G, false, false, 0, 2>), grid, block, 0, s, p, order);                \
  } while (0)
    switch (p.gather) {
      case TC_GATHER_LINEAR: TC_LAUNCH16_ILV(TC_GATHER_LINEAR); break;
      case TC_GATHER_CONV3x3: TC_LAUNCH16_ILV(TC_GATHER_CONV3x3); break;
      default: TC_LAUNCH16_ILV(TC_GATHER_CONVT3); break;
    }
#undef TC_LAUNCH16_ILV
    return 1;
  }
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
