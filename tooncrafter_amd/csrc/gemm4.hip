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
//   per K-step kb (stage st = kb & 1; slice = 32 of the 64 K values; set = 8 + 8 fragments of a slice in registers):
//     24 MFMAs of slice 0 (set 0) | fragments of slice 1 -> set 1
//     s_waitcnt lgkmcnt(0); s_barrier  (B1)             : every wave has read all of stage st
//     40 MFMAs of slice 0 | 8 requests of step kb + 2 (W rows, into stage st)
//     s_waitcnt vmcnt(8); s_barrier    (B2)             : step kb + 1 has landed in stage st ^ 1, for every wave
//     64 MFMAs of slice 1 (set 1) | fragments of slice 0 of step kb + 1 -> set 0 | 8 requests of step kb + 2 (A rows)
//   A request is waited for more than one whole K-step (2048 matrix cycles) after it was issued.
// LDS: 2 stages x (256 + 256) rows x 128 B = 128 KiB (one block per CU), XOR swizzle (row >> 1) & 7 on the SOURCE chunk of the
// DMA as everywhere (conflict-free ds_read_b128 of the 16x16x32 operand layout from 16-aligned row blocks).
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
constexpr int G4_STAGE = (G4_BM + G4_BN) * TC_BK * 2;        // 64 KiB
constexpr int G4_RSTEP = G4_THREADS / 8;                     // rows per loader pass: 32
constexpr int G4_R = G4_BM / G4_RSTEP;                       // loader passes over the A rows (and over the W rows): 8
constexpr int G4_PIECE = G4_RSTEP * TC_BK * 2;               // LDS bytes from one loader pass to the next: 4 KiB
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

template <bool GEGLU, int V>
__global__ __launch_bounds__(G4_THREADS) void gemm4_kernel(const TcGemmParams p, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * G4_STAGE];

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

  // ---- loader geometry (gemm16.hip): thread -> (row lrow + 32 q, 16-byte chunk), swizzle on the SOURCE chunk; one wave
  // instruction fills 1 KiB = 8 rows
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  uint32_t a_voff[G4_R], b_voff[G4_R];
#pragma unroll
  for (int q = 0; q < G4_R; ++q) {
    const int ml = lrow + G4_RSTEP * q;
    a_voff[q] = tile_m * G4_BM + ml < p.m ? (uint32_t)((int64_t)ml * p.lda * 2 + chunk * 16) : TC_OOB;
    const int n = tile_n * G4_BN + ml;
    b_voff[q] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const int nk = (p.k + TC_BK - 1) / TC_BK;
  const bool k_ragged = (p.k & (TC_BK - 1)) != 0;

  uint32_t r_soff = 0, r_dst = 0;
  auto prep = [&](int kb, int stage) {
    const int k0 = kb * TC_BK;
    if (k_ragged && kb == nk - 1) {                       // the K tail: its chunks are zero-filled by an out-of-range offset; only
      const uint32_t kill = (k0 + chunk * 8 >= p.k) ? TC_OOB : 0u;      // the tile's LAST requests see it (chunk = the SOURCE chunk)
#pragma unroll
      for (int q = 0; q < G4_R; ++q) { a_voff[q] |= kill; b_voff[q] |= kill; }
    }
    r_soff = (uint32_t)k0 * 2u;
    r_dst = lds0 + (uint32_t)(stage * G4_STAGE + wave_u * 1024);
  };
  // piece q of the prepared K-step: 0..7 = W row passes, 8..15 = A row passes
  auto issue_piece = [&](auto Q_) {
    constexpr int q = decltype(Q_)::value;
    if constexpr (q < G4_R) g8_dma16(w_srd, r_dst + G4_BM * TC_BK * 2 + q * G4_PIECE, b_voff[q], r_soff);
    else g8_dma16(a_srd, r_dst + (q - G4_R) * G4_PIECE, a_voff[q - G4_R], r_soff);
  };
  auto issue_all = [&]() {
    issue_piece(ic<0>{}); issue_piece(ic<1>{}); issue_piece(ic<2>{}); issue_piece(ic<3>{});
    issue_piece(ic<4>{}); issue_piece(ic<5>{}); issue_piece(ic<6>{}); issue_piece(ic<7>{});
    issue_piece(ic<8>{}); issue_piece(ic<9>{}); issue_piece(ic<10>{}); issue_piece(ic<11>{});
    issue_piece(ic<12>{}); issue_piece(ic<13>{}); issue_piece(ic<14>{}); issue_piece(ic<15>{});
  };

  f32x4_t acc[G4_NT][G4_NT];
#pragma unroll
  for (int i = 0; i < G4_NT; ++i)
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // v_mfma_f32_16x16x32_bf16 operands: lane holds row (lane & 15) of its 16-row tile, k = 8 (lane >> 4) .. +7 of the 32-deep
  // slice -> one ds_read_b128 at 16-byte chunk 4 ks + (lane >> 4) of that row; tile rows step by 16, so (row >> 1) & 7 is the lane's
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int a_row0 = (wm * G4_WT + frow) * (TC_BK * 2);
  const int b_row0 = G4_BM * TC_BK * 2 + (wn * G4_WT + frow) * (TC_BK * 2);
  const int f_sw = (frow >> 1) & 7;                          // wm * 128 and wn * 128 are multiples of 16: they do not enter
  auto read_frags = [&](int stage, int ks, bf16x8 (&af)[G4_NT], bf16x8 (&bf)[G4_NT]) {
    const char* s0 = smem + stage * G4_STAGE + (((ks * 4 + fq) ^ f_sw) << 4);
#pragma unroll
    for (int i = 0; i < G4_NT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(s0 + a_row0 + i * (16 * TC_BK * 2));
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(s0 + b_row0 + j * (16 * TC_BK * 2));
  };
  auto mma_row = [&](auto I_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT]) {
    constexpr int i = decltype(I_)::value;
#pragma unroll
    for (int j = 0; j < G4_NT; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
  };

  // ---- main loop
  prep(0, 0);
  issue_all();
  if (nk > 1) {
    prep(1, 1);
    issue_all();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // step 0 has landed, the 16 requests of step 1 may be in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  g8_barrier();
  bf16x8 af0[G4_NT], bf0[G4_NT], af1[G4_NT], bf1[G4_NT];
  read_frags(0, 0, af0, bf0);
  // one K-step; ISSUE: the 16 requests of step kb + 2 go out between the MFMA rows of the second slice.  Two loops (all steps
  // but the last two issue) instead of a run-time condition around every pair of requests: no branches inside a step
  // MFMA row i of a slice with the requests (if any) of pieces Q0, Q1 behind its fourth and eighth MFMA: six scalar /
  // vector-memory instructions each, issued in the shadow of the MFMA in flight.  MFMAs and requests are both volatile asm:
  // this IS the instruction order.
  auto row = [&](auto I_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT], auto Q0_, auto Q1_) {
    constexpr int i = decltype(I_)::value, q0 = decltype(Q0_)::value, q1 = decltype(Q1_)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    if constexpr (q0 >= 0) issue_piece(ic<(q0 >= 0 ? q0 : 0)>{});
#pragma unroll
    for (int j = 4; j < 8; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    if constexpr (q1 >= 0) issue_piece(ic<(q1 >= 0 ? q1 : 0)>{});
  };
  // One K-step, TWO barriers.  B1 sits where every wave has the step's last fragments in registers: from there on stage st is
  // free and the requests of step kb + 2 go out -- 8 before B2, 8 after it -- so a request is waited for MORE than a whole K-step
  // after it was issued (with a single barrier at mid-step the last request had half a step: 1024 matrix cycles, less than an
  // L2 round trip under load).  B2 = step kb + 1 has landed for every wave: the counted wait leaves the 8 newest requests in flight.
  auto kstep_v0 = [&](auto ISSUE_, int kb) {
    constexpr bool ISSUE = decltype(ISSUE_)::value != 0;
    constexpr int N = -1;
    const int st = kb & 1;
    row(ic<0>{}, af0, bf0, ic<N>{}, ic<N>{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(st, 1, af1, bf1);                             // behind the first MFMA row: hipcc's waits in front of that row only see old reads
    __builtin_amdgcn_sched_barrier(0);
    row(ic<1>{}, af0, bf0, ic<N>{}, ic<N>{});
    row(ic<2>{}, af0, bf0, ic<N>{}, ic<N>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_barrier();                                            // B1: stage st has been read by every wave
    if constexpr (ISSUE) {
      prep(kb + 2, st);
      row(ic<3>{}, af0, bf0, ic<0>{}, ic<1>{});
      row(ic<4>{}, af0, bf0, ic<2>{}, ic<3>{});
      row(ic<5>{}, af0, bf0, ic<4>{}, ic<5>{});
      row(ic<6>{}, af0, bf0, ic<6>{}, ic<7>{});
      row(ic<7>{}, af0, bf0, ic<N>{}, ic<N>{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      row(ic<3>{}, af0, bf0, ic<N>{}, ic<N>{}); row(ic<4>{}, af0, bf0, ic<N>{}, ic<N>{}); row(ic<5>{}, af0, bf0, ic<N>{}, ic<N>{});
      row(ic<6>{}, af0, bf0, ic<N>{}, ic<N>{}); row(ic<7>{}, af0, bf0, ic<N>{}, ic<N>{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    g8_barrier();                                            // B2: step kb + 1 is complete in stage st ^ 1
    row(ic<0>{}, af1, bf1, ic<N>{}, ic<N>{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(st ^ 1, 0, af0, bf0);                         // (in the last step it reads a dead stage, nothing uses it)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ISSUE) {
      row(ic<1>{}, af1, bf1, ic<8>{}, ic<9>{});
      row(ic<2>{}, af1, bf1, ic<10>{}, ic<11>{});
      row(ic<3>{}, af1, bf1, ic<12>{}, ic<13>{});
      row(ic<4>{}, af1, bf1, ic<14>{}, ic<15>{});
    } else {
      row(ic<1>{}, af1, bf1, ic<N>{}, ic<N>{}); row(ic<2>{}, af1, bf1, ic<N>{}, ic<N>{});
      row(ic<3>{}, af1, bf1, ic<N>{}, ic<N>{}); row(ic<4>{}, af1, bf1, ic<N>{}, ic<N>{});
    }
    row(ic<5>{}, af1, bf1, ic<N>{}, ic<N>{}); row(ic<6>{}, af1, bf1, ic<N>{}, ic<N>{}); row(ic<7>{}, af1, bf1, ic<N>{}, ic<N>{});
  };
  // ---- the same K-step with the fragment reads SPREAD between the MFMAs.  Counters of the burst version (profiles/
  // r05_pmc_gemm8_gemm4_hipblaslt_8192.txt: matrix pipe 53 % busy where the library's kernel of the same tile keeps it 86 % busy)
  // point at the bursts: four waves leave a barrier together and each issues 16 ds_read_b128 -- 64 KiB, 512 cycles of the CU's LDS
  // pipe -- in front of its next MFMAs; instruction issue is in order, so the matrix pipe runs dry behind the eight MFMAs queued
  // before the burst.  One read per four MFMAs keeps the LDS pipe half busy all the time and never stands in front of an MFMA.
  // half-row h of MFMA row i: four MFMAs
  auto quad = [&](auto I_, auto H_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT]) {
    constexpr int i = decltype(I_)::value, h0 = decltype(H_)::value * 4;
#pragma unroll
    for (int j = h0; j < h0 + 4; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
  };
  // fragment f of the next slice: 0..7 = W tiles (all needed by the next slice's first MFMA row), 8..15 = A tiles (tile i by row i)
  auto read_one = [&](auto F_, int stage, int ks, bf16x8 (&af)[G4_NT], bf16x8 (&bf)[G4_NT]) {
    constexpr int f = decltype(F_)::value;
    const char* s0 = smem + stage * G4_STAGE + (((ks * 4 + fq) ^ f_sw) << 4);
    if constexpr (f < G4_NT) bf[f] = *reinterpret_cast<const bf16x8*>(s0 + b_row0 + f * (16 * TC_BK * 2));
    else af[f - G4_NT] = *reinterpret_cast<const bf16x8*>(s0 + a_row0 + (f - G4_NT) * (16 * TC_BK * 2));
  };
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
  // one MFMA row of the CURRENT slice (fragments af / bf) with two fragment reads of the NEXT slice (F0, F1; -1 = none) and up to
  // two requests (Q0, Q1; -1 = none): [4 MFMA] read F0, request Q0 [4 MFMA] read F1, request Q1
  auto mrow = [&](auto I_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT], auto F0_, auto F1_, int nstage, int nks,
                  bf16x8 (&afn)[G4_NT], bf16x8 (&bfn)[G4_NT], auto Q0_, auto Q1_) {
    constexpr int f0 = decltype(F0_)::value, f1 = decltype(F1_)::value, q0 = decltype(Q0_)::value, q1 = decltype(Q1_)::value;
    quad(I_, ic<0>{}, af, bf);
    G4_SB();
    if constexpr (f0 >= 0) read_one(ic<(f0 >= 0 ? f0 : 0)>{}, nstage, nks, afn, bfn);
    if constexpr (q0 >= 0) issue_piece(ic<(q0 >= 0 ? q0 : 0)>{});
    G4_SB();
    quad(I_, ic<1>{}, af, bf);
    G4_SB();
    if constexpr (f1 >= 0) read_one(ic<(f1 >= 0 ? f1 : 0)>{}, nstage, nks, afn, bfn);
    if constexpr (q1 >= 0) issue_piece(ic<(q1 >= 0 ? q1 : 0)>{});
    G4_SB();
  };
  // V = 1: ONE barrier per K-step, at mid-step.  Slice 0's MFMAs carry the reads of slice 1 (same stage), slice 1's MFMAs the
  // reads of the next step's slice 0 and the 16 requests of step kb + 2.  Reads end one MFMA row before the barrier.
  auto kstep_v1 = [&](auto ISSUE_, int kb) {
    constexpr bool ISSUE = decltype(ISSUE_)::value != 0;
    constexpr int N = -1;
    const int st = kb & 1;
    mrow(ic<0>{}, af0, bf0, ic<0>{}, ic<1>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<1>{}, af0, bf0, ic<2>{}, ic<3>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<2>{}, af0, bf0, ic<4>{}, ic<5>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<3>{}, af0, bf0, ic<6>{}, ic<7>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<4>{}, af0, bf0, ic<8>{}, ic<9>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<5>{}, af0, bf0, ic<10>{}, ic<11>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    mrow(ic<6>{}, af0, bf0, ic<12>{}, ic<13>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    quad(ic<7>{}, ic<0>{}, af0, bf0);
    G4_SB();
    read_one(ic<14>{}, st, 1, af1, bf1);
    read_one(ic<15>{}, st, 1, af1, bf1);
    G4_SB();
    quad(ic<7>{}, ic<1>{}, af0, bf0);
    G4_SB();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    g8_barrier();                                            // stage st is free, stage st ^ 1 is complete, for every wave
    if constexpr (ISSUE) {
      prep(kb + 2, st);
      mrow(ic<0>{}, af1, bf1, ic<0>{}, ic<1>{}, st ^ 1, 0, af0, bf0, ic<0>{}, ic<1>{});
      mrow(ic<1>{}, af1, bf1, ic<2>{}, ic<3>{}, st ^ 1, 0, af0, bf0, ic<2>{}, ic<3>{});
      mrow(ic<2>{}, af1, bf1, ic<4>{}, ic<5>{}, st ^ 1, 0, af0, bf0, ic<4>{}, ic<5>{});
      mrow(ic<3>{}, af1, bf1, ic<6>{}, ic<7>{}, st ^ 1, 0, af0, bf0, ic<6>{}, ic<7>{});
      mrow(ic<4>{}, af1, bf1, ic<8>{}, ic<9>{}, st ^ 1, 0, af0, bf0, ic<8>{}, ic<9>{});
      mrow(ic<5>{}, af1, bf1, ic<10>{}, ic<11>{}, st ^ 1, 0, af0, bf0, ic<10>{}, ic<11>{});
      mrow(ic<6>{}, af1, bf1, ic<12>{}, ic<13>{}, st ^ 1, 0, af0, bf0, ic<12>{}, ic<13>{});
      mrow(ic<7>{}, af1, bf1, ic<14>{}, ic<15>{}, st ^ 1, 0, af0, bf0, ic<14>{}, ic<15>{});
    } else {
      mrow(ic<0>{}, af1, bf1, ic<0>{}, ic<1>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<1>{}, af1, bf1, ic<2>{}, ic<3>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<2>{}, af1, bf1, ic<4>{}, ic<5>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<3>{}, af1, bf1, ic<6>{}, ic<7>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<4>{}, af1, bf1, ic<8>{}, ic<9>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<5>{}, af1, bf1, ic<10>{}, ic<11>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<6>{}, af1, bf1, ic<12>{}, ic<13>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<7>{}, af1, bf1, ic<14>{}, ic<15>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
    }
  };
  // V = 2: TWO barriers.  The 16 reads of slice 1 ride on the first four MFMA rows (one per two MFMAs), B1 behind the fifth row frees
  // stage st, the requests of step kb + 2 start there (a whole K-step and more before they are waited for), B2 at mid-step.
  auto drow = [&](auto I_, const bf16x8 (&af)[G4_NT], const bf16x8 (&bf)[G4_NT], auto FB_, int nstage, int nks, bf16x8 (&afn)[G4_NT],
                  bf16x8 (&bfn)[G4_NT]) {                      // an MFMA row with FOUR reads (fragments FB .. FB + 3)
    constexpr int i = decltype(I_)::value, fb = decltype(FB_)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    G4_SB(); read_one(ic<fb>{}, nstage, nks, afn, bfn); G4_SB();
#pragma unroll
    for (int j = 2; j < 4; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    G4_SB(); read_one(ic<fb + 1>{}, nstage, nks, afn, bfn); G4_SB();
#pragma unroll
    for (int j = 4; j < 6; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    G4_SB(); read_one(ic<fb + 2>{}, nstage, nks, afn, bfn); G4_SB();
#pragma unroll
    for (int j = 6; j < 8; ++j) g4_mfma(acc[i][j], af[i], bf[j]);
    G4_SB(); read_one(ic<fb + 3>{}, nstage, nks, afn, bfn); G4_SB();
  };
  auto kstep_v2 = [&](auto ISSUE_, int kb) {
    constexpr bool ISSUE = decltype(ISSUE_)::value != 0;
    constexpr int N = -1;
    const int st = kb & 1;
    drow(ic<0>{}, af0, bf0, ic<0>{}, st, 1, af1, bf1);
    drow(ic<1>{}, af0, bf0, ic<4>{}, st, 1, af1, bf1);
    drow(ic<2>{}, af0, bf0, ic<8>{}, st, 1, af1, bf1);
    drow(ic<3>{}, af0, bf0, ic<12>{}, st, 1, af1, bf1);
    mrow(ic<4>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_barrier();                                            // B1: stage st has been read by every wave
    if constexpr (ISSUE) {
      prep(kb + 2, st);
      mrow(ic<5>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<0>{}, ic<1>{});
      mrow(ic<6>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<2>{}, ic<3>{});
      mrow(ic<7>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<4>{}, ic<5>{});
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // step kb + 1 has landed: the six newest requests stay in flight
    } else {
      mrow(ic<5>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
      mrow(ic<6>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
      mrow(ic<7>{}, af0, bf0, ic<N>{}, ic<N>{}, st, 1, af1, bf1, ic<N>{}, ic<N>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    g8_barrier();                                            // B2: step kb + 1 is complete in stage st ^ 1
    if constexpr (ISSUE) {
      mrow(ic<0>{}, af1, bf1, ic<0>{}, ic<1>{}, st ^ 1, 0, af0, bf0, ic<6>{}, ic<7>{});
      mrow(ic<1>{}, af1, bf1, ic<2>{}, ic<3>{}, st ^ 1, 0, af0, bf0, ic<8>{}, ic<9>{});
      mrow(ic<2>{}, af1, bf1, ic<4>{}, ic<5>{}, st ^ 1, 0, af0, bf0, ic<10>{}, ic<11>{});
      mrow(ic<3>{}, af1, bf1, ic<6>{}, ic<7>{}, st ^ 1, 0, af0, bf0, ic<12>{}, ic<13>{});
      mrow(ic<4>{}, af1, bf1, ic<8>{}, ic<9>{}, st ^ 1, 0, af0, bf0, ic<14>{}, ic<15>{});
    } else {
      mrow(ic<0>{}, af1, bf1, ic<0>{}, ic<1>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<1>{}, af1, bf1, ic<2>{}, ic<3>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<2>{}, af1, bf1, ic<4>{}, ic<5>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<3>{}, af1, bf1, ic<6>{}, ic<7>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
      mrow(ic<4>{}, af1, bf1, ic<8>{}, ic<9>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
    }
    mrow(ic<5>{}, af1, bf1, ic<10>{}, ic<11>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
    mrow(ic<6>{}, af1, bf1, ic<12>{}, ic<13>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
    mrow(ic<7>{}, af1, bf1, ic<14>{}, ic<15>{}, st ^ 1, 0, af0, bf0, ic<N>{}, ic<N>{});
  };
#undef G4_SB
  auto kstep = [&](auto ISSUE_, int kb) {
    if constexpr (V == 0) kstep_v0(ISSUE_, kb);
    else if constexpr (V == 1) kstep_v1(ISSUE_, kb);
    else kstep_v2(ISSUE_, kb);
  };
  int kb = 0;
  for (; kb + 2 < nk; ++kb) kstep(ic<1>{}, kb);
  for (; kb < nk; ++kb) kstep(ic<0>{}, kb);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                           // the epilogue slabs reuse the stage buffers

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
  if (p.gather != TC_GATHER_LINEAR || p.gn_part || p.a_norm || (p.n & 7) || p.k < 2 * TC_BK) return 0;
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
  const int v = [] { const char* e = getenv("TC_G4_VARIANT"); return e ? atoi(e) : 2; }();      // K-loop variant (A/B runs): see kstep_v0 / v1 / v2
#define TC_G4_LAUNCH(V_)                                                                           \
  do {                                                                                             \
    if (geglu) hipLaunchKernelGGL((gemm4_kernel<true, V_>), grid, block, 0, s, p, (int)total);     \
    else hipLaunchKernelGGL((gemm4_kernel<false, V_>), grid, block, 0, s, p, (int)total);          \
  } while (0)
  if (v == 0) TC_G4_LAUNCH(0);
  else if (v == 1) TC_G4_LAUNCH(1);
  else TC_G4_LAUNCH(2);
#undef TC_G4_LAUNCH
  return 1;
}
