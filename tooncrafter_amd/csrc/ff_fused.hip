// Fused feed-forward of a level-0 transformer block, gfx950:
//
//     out = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2            x: [M, 320] bf16, hidden 1280
//
// (reference lvdm/modules/attention.py:415-442 FeedForward / GEGLU behind norm3, attention.py:244-246).  As three launches
// -- LayerNorm, the GEGLU projection (81920 x 2560 x 320) and ff2 (81920 x 320 x 1280) -- the block writes its 210 MB
// hidden tensor to HBM and reads it back: 240 + 111 (+ 20) us, both GEMMs far from the matrix roof because they are bound by
// those bytes and by the GEGLU epilogue (profiles/r03_ablate_gemm.txt, r04_gemm8_ablation.txt).  Here the hidden tensor
// never exists:
//
//  * a block owns 128 rows; wave (wm, wn) of its 8 waves owns rows wm*32..+32.  The rows' LayerNorm runs in REGISTERS on
//    the MFMA A-operand layout (a row's 320 values sit in two lanes: one lane swap) and the normalised rows stay there as
//    the A fragments of the first product for the whole tile (80 VGPRs) -- x is read once, nothing of it goes through LDS;
//  * the hidden dimension is walked in 20 chunks of 64.  Per chunk: ff1 = 5 K-steps against a [128 packed rows x 64]
//    W1 tile (wave: its 32 rows x 64 packed columns = one value block + one gate block, 8 MFMAs per K-step; the stage
//    holds, per 64 rows, [32 values | 32 gates], so a value and its gate meet in the same lane and register); GEGLU on the
//    accumulators; the 32 x 32 hidden values of the wave go to LDS as bf16 in A-operand layout; ff2 = the tile's
//    [128 x 64] hidden chunk against W2's [320 x 64] slice into the wave's 32 x 160 output accumulators (80 VGPRs,
//    20 MFMAs), which live across all 20 chunks;
//  * the weights (2.4 MB, L2-resident) stream through LDS by LDS-DMA from inline asm: W1 K-tiles through a ring of four
//    16 KiB stages (requested two steps ahead), W2's 40 KiB slice once per chunk (five pieces spread over the chunk's
//    steps).  The stream does not depend on the tile, so it simply runs on across the tiles of a persistent block;
//  * the two wave GROUPS (wm >> 1: one wave per SIMD each) run one barrier interval apart, as in gemm8.hip: while one
//    group issues its MFMAs the other reads its fragments / requests tiles / evaluates GEGLU.
//
// LDS: W1 ring 64 KiB | W2 slice 40 KiB | hidden chunk 2 x 16 KiB (also the epilogue slabs) | biases 12 KiB = 148 KiB.
// Order of the fp32 sums: ff1 over K in ascending 16-slices, ff2 over the hidden index in ascending 16-slices -- the
// order of the tiled kernels, so the result equals LayerNorm -> tc_gemm_bf16(GEGLU) -> tc_gemm_bf16(+residual) up to the
// LayerNorm's own rounding.
#include "gemm_common.h"
#include "gemm_persist.h"

#include <stdlib.h>

namespace {

constexpr int FF_C = 320, FF_H = 1280, FF_BM = 128, FF_THREADS = 512;
constexpr int FF_CH = 64;                       // hidden columns per chunk = 128 packed W1 rows
constexpr int FF_NCH = FF_H / FF_CH;            // 20 chunks
constexpr int FF_KT = FF_C / TC_BK;             // 5 K-steps of ff1 per chunk
constexpr int FF_W1_STAGE = 128 * 128;          // 16 KiB: 128 packed rows x 64 k
constexpr int FF_W1_OFF = 0;                    // ring of 4
constexpr int FF_W2_OFF = 4 * FF_W1_STAGE;      // 40 KiB: 320 rows x 64 k
constexpr int FF_W2_BYTES = 320 * 128;
constexpr int FF_H_OFF = FF_W2_OFF + FF_W2_BYTES;   // 2 x 16 KiB: 128 rows x 64 hidden
constexpr int FF_H_BYTES = 128 * 128;
constexpr int FF_B_OFF = FF_H_OFF + 2 * FF_H_BYTES; // b1 (2560 fp32) | b2 (320 fp32)
constexpr int FF_P_OFF = FF_B_OFF + (2 * FF_H + FF_C) * 4;   // FF_NPARK parked A fragments: [slot][wm][lane] x 16 B
constexpr int FF_NPARK = 3;                     // the last K-slices of the normalised rows live in LDS, not in registers
constexpr int FF_NRES = 20 - FF_NPARK;
constexpr int FF_LDS = FF_P_OFF + FF_NPARK * 4096;
static_assert(FF_LDS <= 160 * 1024, "LDS");

struct FfArgs {
  const bf16_t* x; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; bf16_t* out;
  int m, ldx, ldo, ln;
  float eps;
  int tiles;
  unsigned long long* trace;     // TC_FF_TRACE (ABL bit 16 build): s_memtime after every barrier of block 0's waves 0 and 4
};

template <int N>
__device__ __forceinline__ void ff_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LA: W1 K-tiles requested ahead (2 or 3 of the ring's 4 stages; 4 = three ahead with the requests issued in the MFMA segment,
// between the MFMAs, instead of in the read segment: 274-280 us against 265-268, profiles/r04_ff_fused_bench.txt -- kept as a switch).  ABL: timing ablations (TC_FF_ABLATE; wrong results) --
// 1 no GELU arithmetic, 2 no weight requests, 4 no ff1 MFMAs, 8 no ff2 MFMAs.
template <int LA, int ABL, int GI>
__global__ __launch_bounds__(FF_THREADS, 2) void ff_fused_kernel(const FfArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[FF_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int grp = wave_u >> 2;                   // = wm >> 1: waves w and w + 4 share a SIMD, one of each group
  const int frow = lane & 31, fhalf = lane >> 5;

  const g8_srd_t w1_srd = g8_make_srd(p.w1, (int64_t)2 * FF_H * FF_C * 2);
  const g8_srd_t w2_srd = g8_make_srd(p.w2, (int64_t)FF_C * FF_H * 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // biases into LDS once (no global load may sit inside the chunk loop: hipcc would wait vmcnt(0) for it and drain the stream)
  {
    float* bl = reinterpret_cast<float*>(smem + FF_B_OFF);
    for (int i = tid; i < 2 * FF_H; i += FF_THREADS) bl[i] = p.b1[i];
    for (int i = tid; i < FF_C; i += FF_THREADS) bl[2 * FF_H + i] = p.b2[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // ... and in LDS before the first barrier
  }

  // ---- weight stream: thread -> (row lrow + 64 i, 16-byte chunk) of a 64-row pass; the XOR swizzle on the SOURCE chunk
  const int lrow = tid >> 3;
  const int sch = (tid & 7) ^ ((lrow >> 1) & 7);
  // W1 arrives in the GEMM's GEGLU packing (lvdm/common.py pack_geglu: every 32 rows = 16 value rows | their 16 gate rows);
  // the stage wants, per 64 rows, [32 values | 32 gates], so that a value and its gate meet in one lane of the two 32 x 32
  // accumulator blocks -- a permutation of the SOURCE row, free in the request's address
  const int w1_src = ((lrow & 31) >> 4) * 32 + (lrow >> 5) * 16 + (lrow & 15);
  const uint32_t v1 = (uint32_t)(w1_src * FF_C * 2 + sch * 16);       // W1: + soffset (packed row block, K-step, pass)
  const uint32_t v2 = (uint32_t)(lrow * FF_H * 2 + sch * 16);         // W2: + soffset (hidden chunk, pass)
  const uint32_t dma_dst = lds0 + wave_u * 1024;
  auto dma_w1 = [&](int q) {                                          // W1 K-tile q of the cyclic stream -> ring stage q & 3
    if constexpr (ABL & 2) return;
    const int qq = q % (FF_NCH * FF_KT);
    const int c = qq / FF_KT, kt = qq - c * FF_KT;
    const uint32_t so = (uint32_t)((c * 128 * FF_C + kt * TC_BK) * 2);
    const uint32_t dst = dma_dst + FF_W1_OFF + (q & 3) * FF_W1_STAGE;
    g8_dma16(w1_srd, dst, v1, so);
    g8_dma16(w1_srd, dst + 8192, v1, so + 64 * FF_C * 2);
  };
  auto dma_w1_half = [&](int q, int half) {                           // one of the K-tile's two pieces (LA == 4)
    if constexpr (ABL & 2) return;
    const int qq = q % (FF_NCH * FF_KT);
    const int c = qq / FF_KT, kt = qq - c * FF_KT;
    const uint32_t so = (uint32_t)((c * 128 * FF_C + kt * TC_BK) * 2) + (uint32_t)half * (64 * FF_C * 2);
    g8_dma16(w1_srd, dma_dst + FF_W1_OFF + (q & 3) * FF_W1_STAGE + half * 8192, v1, so);
  };
  auto dma_w2 = [&](int c, int piece) {                               // rows 64 piece .. +64 of W2's slice for hidden chunk c
    if constexpr (ABL & 2) return;
    const int cc = c % FF_NCH;
    const uint32_t so = (uint32_t)((piece * 64 * FF_H + cc * FF_CH) * 2);
    g8_dma16(w2_srd, dma_dst + FF_W2_OFF + piece * 8192, v2, so);
  };

  // ---- fragment addressing (32x32x16 MFMA): lane holds row lane & 31, k = 8 (lane >> 5) .. of slice kk -> chunk 2 kk + (lane >> 5)
  // (computed where they are used, from the lane id: the kernel lives at the register limit -- 80 A-fragment + 80 output +
  // 32 ff1 accumulator + 32 B-fragment registers -- and every address kept across the chunk loop is a spill into scratch,
  // i.e. a vector-memory load inside the counted LDS-DMA stream)
  auto coff = [&](int kk) { return ((kk * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4; };
  auto w1_row = [&]() { return (wn * 64 + frow) * 128; };          // + j * 4096 (value / gate block)
  auto h_row = [&]() { return (wm * 32 + frow) * 128; };
  auto w2_row = [&]() { return (wn * 160 + frow) * 128; };         // + j * 4096

  // the tile's normalised rows, A fragments of ff1: K-slices 0 .. FF_NRES-1 in registers, the last FF_NPARK in LDS (both
  // N-waves of a row group write the same bytes).  80 + 80 + 32 + 32 registers of fragments and accumulators do not
  // leave room for addressing under the 256 a wave of an 8-wave block gets; the parked slices are read in the MFMA
  // segment of the last K-step into B-fragment registers its first MFMAs have just released.
  bf16x8 xa[FF_NRES];
  char* const park = smem + FF_P_OFF + wm * 1024 + lane * 16;
  f32x16 out_acc[5];
  f32x16 acc_v, acc_g;

  // ABL bit 16: per-interval timing.  Block 0, waves 0 (group 0) and 4 (group 1), second tile, first three chunks: the
  // shader clock after every barrier -> trace[wave >> 2][n]
  int tr_n = 0;
  bool tr_on = false;
  auto bar = [&]() {
    g8_barrier();
    if constexpr (ABL & 16) {
      if (tr_on && tr_n < 64) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) p.trace[(wave_u >> 2) * 64 + tr_n] = t;
        ++tr_n;
      }
    }
  };
  int q = 0;                                          // W1 K-tile stream position consumed next (cyclic)
  int cw2 = 0;                                        // hidden chunk whose W2 slice is requested next (cyclic): the one
                                                      // the CURRENT chunk's ff2 reads -- requested during its ff1 steps
  // prologue of the stream: W1(0), W1(1)
  dma_w1(0);
  dma_w1(1);
  if constexpr (LA >= 3) dma_w1(2);
  ff_wait_vmcnt<0>();
  g8_barrier();

  const float* bl = reinterpret_cast<const float*>(smem + FF_B_OFF);

  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    // ---- the wave's 32 rows: raw -> LayerNorm -> A fragments
    const int row = tile * FF_BM + wm * 32 + frow;
    const bool rok = row < p.m;
    {
      u32x4 raw[20];
      const bf16_t* xr = p.x + (int64_t)(rok ? row : p.m - 1) * p.ldx + 8 * fhalf;       // rows past M: a valid row, never stored
#pragma unroll
      for (int s = 0; s < 20; ++s) raw[s] = *reinterpret_cast<const u32x4*>(xr + 16 * s);
      if (p.ln) {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 20; ++s) {
          float f[8];
          unpack8(raw[s], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += f[e];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / FF_C);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < 20; ++s) {
          float f[8];
          unpack8(raw[s], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq += d * d; }
        }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / FF_C) + p.eps);
#pragma unroll
        for (int s = 0; s < 20; ++s) {
          float f[8];
          unpack8(raw[s], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd;
          raw[s] = pack8(f);
        }
      }
#pragma unroll
      for (int s = 0; s < 20; ++s) {
        if (s < FF_NRES) xa[s] = __builtin_bit_cast(bf16x8, raw[s]);
        else *reinterpret_cast<u32x4*>(park + (s - FF_NRES) * 4096) = raw[s];
      }
      // without LayerNorm nothing has consumed the row loads yet: a load still "pending" at the chunk loop's header makes
      // the compiler wait vmcnt(0) at its first use INSIDE the loop, every chunk -- so they are consumed here, once
#pragma unroll
      for (int s = 0; s < FF_NRES; ++s) asm volatile("" ::"v"(xa[s]));
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out_acc[j][r] = 0.f;

    // the stagger, per tile: group 1 runs one barrier interval behind group 0 through the chunks (and is let catch up
    // before the epilogue, so that the two groups' epilogues and row loads -- long, barrier-free -- run side by side)
    if (grp == 1) g8_barrier();
    for (int c = 0; c < FF_NCH; ++c) {
      if constexpr (ABL & 16) tr_on = blockIdx.x == 0 && (wave_u & 3) == 0 && tile == (int)(blockIdx.x + gridDim.x) && c < 3;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_v[r] = 0.f; acc_g[r] = 0.f; }
      // ---- ff1: five K-steps; step s reads W1 K-tile q (ring stage q & 3) and requests K-tile q + 2
      auto ff1_step = [&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const char* st = smem + FF_W1_OFF + (q & 3) * FF_W1_STAGE + w1_row();
        bf16x8 bw[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) bw[j][kk] = *reinterpret_cast<const bf16x8*>(st + j * 4096 + coff(kk));
        if constexpr (LA != 4) {
          dma_w1(q + LA);
          if (s == 1) { dma_w2(cw2, 0); dma_w2(cw2, 1); }
          if (s >= 2) { dma_w2(cw2, s); }
        }
        // the K-tile the NEXT step reads has landed (this wave's pieces; the barrier makes it everybody's): the counts are
        // the requests issued after it -- see the table above the steps
        if constexpr (LA == 2) {
          if (s == 0) ff_wait_vmcnt<2>();
          else if (s == 2) ff_wait_vmcnt<5>();
          else ff_wait_vmcnt<4>();
        } else if constexpr (LA == 3) {
          if (s == 0) ff_wait_vmcnt<2>();
          else if (s == 1) ff_wait_vmcnt<6>();
          else if (s == 3) ff_wait_vmcnt<8>();
          else ff_wait_vmcnt<7>();
        } else {
          // requests in the MFMA segments: M_s issues W1(q + 3) (two pieces), then piece s of this chunk's W2 slice.  The
          // tile the next step reads was requested two MFMA segments ago, first there; behind it: that segment's W2 piece
          // and the last segment's three requests (across the chunk seam everything was drained at the end of G)
          if (s == 0) ff_wait_vmcnt<0>();
          else if (s == 1) ff_wait_vmcnt<3>();
          else ff_wait_vmcnt<4>();
        }
        bar();
        __builtin_amdgcn_s_setprio(1);
        auto mm = [&](auto KK_) {
          constexpr int kk = decltype(KK_)::value, ks = 4 * s + kk;
          if constexpr (ABL & 4) {
            asm volatile("" : "+v"(acc_v), "+v"(acc_g) : "v"(bw[0][kk]), "v"(bw[1][kk]));
          } else if constexpr (ks < FF_NRES) {
            acc_v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks], bw[0][kk], acc_v, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks], bw[1][kk], acc_g, 0, 0, 0);
          } else {
            const bf16x8 pa = *reinterpret_cast<const bf16x8*>(park + (ks - FF_NRES) * 4096);
            acc_v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, bw[0][kk], acc_v, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, bw[1][kk], acc_g, 0, 0, 0);
          }
          if constexpr (s == 4) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (LA == 4) {
          mm(ic<0>{});
          __builtin_amdgcn_sched_barrier(0);
          dma_w1_half(q + 3, 0);
          __builtin_amdgcn_sched_barrier(0);
          mm(ic<1>{});
          __builtin_amdgcn_sched_barrier(0);
          dma_w1_half(q + 3, 1);
          __builtin_amdgcn_sched_barrier(0);
          mm(ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          dma_w2(cw2, s);
          __builtin_amdgcn_sched_barrier(0);
          mm(ic<3>{});
        } else {
          mm(ic<0>{});
          mm(ic<1>{});
          mm(ic<2>{});
          mm(ic<3>{});
        }
        __builtin_amdgcn_s_setprio(0);
        bar();
        ++q;
      };
      // Requests in program order, per chunk (W1 = two pieces per thread, a W2 piece = one; the W2 slice requested is THIS
      // chunk's -- its buffer is single, and the other group reads the previous slice until the end of this group's R0,
      // hence nothing for it in R0):
      //     R0: W1(q+2) | R1: W1, p0, p1 | R2: W1, p2 | R3: W1, p3 | R4: W1, p4 | G: none, then vmcnt(0)
      // The drain at the end of G (behind ~600 cycles of GEGLU arithmetic: what is in flight was requested at least an
      // MFMA segment earlier) retires the whole W2 slice and the two K-tiles requested ahead.  The wait at the end of R_s
      // retires W1 K-tile q+1; what may stay in flight is what was requested after it:
      //     s = 0: W1 = 2 | s = 1: W1, p0, p1 = 4 | s = 2: p0, p1, W1, p2 = 5 | s = 3: p2, W1, p3 = 4 | s = 4: p3, W1, p4 = 4
      ff1_step(ic<0>{});
      ff1_step(ic<1>{});
      ff1_step(ic<2>{});
      ff1_step(ic<3>{});
      ff1_step(ic<4>{});
      // ---- G: GEGLU on the accumulators -> the wave's 32 x 32 hidden values, bf16, into the chunk buffer (A layout of ff2)
      {
        // the lane id is taken afresh (and opaquely): an address hoisted out of the chunk loop is a register the loop does
        // not have, i.e. a scratch reload, i.e. a compiler "s_waitcnt vmcnt(0)" at the top of every G draining the stream
        int gl = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(gl));
        const int frow = gl & 31, fhalf = gl >> 5;
        const int pr = c * 128 + wn * 64 + (frow >> 4) * 32 + (frow & 15);   // pack_geglu row of this lane's value; gate 16 on
        const float bv = bl[pr], bg = bl[pr + 16];
        // accumulator register r of a lane is row cr = (r & 3) + 8 (r >> 2) (+ 4 fhalf) of the wave's 32, column frow; its
        // place in the chunk buffer (row-major [128][64] bf16, 16-byte chunks XOR-swizzled by (row >> 1) & 7):
        //     row * 128 + (((hcol >> 3) ^ ((row >> 1) & 7)) << 4) + (hcol & 7) * 2,     hcol = wn * 32 + frow
        // (row >> 1) & 7 = kr | 2 fhalf with kr = (cr >> 1) & 7 in {0, 1, 4, 5}: four lane-dependent bases, the rest immediates
        const int hcol = wn * 32 + frow;
        char* const hrow = smem + FF_H_OFF + (c & 1) * FF_H_BYTES + (wm * 32 + 4 * fhalf) * 128 + (hcol & 7) * 2;
        const int a2 = (hcol >> 3) ^ (2 * fhalf);
        char* const hbase[4] = {hrow + (a2 << 4), hrow + ((a2 ^ 1) << 4), hrow + ((a2 ^ 4) << 4), hrow + ((a2 ^ 5) << 4)};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                             // register pairs: packed fp32 arithmetic
          const int cr = (r & 3) + 8 * (r >> 2);                      // row of r; r + 1 is the next row (r is even)
          const int kr = (cr >> 1) & 7;                               // 0, 1, 4 or 5 -- the same for both
          const tc_f32x2 v = {acc_v[r] + bv, acc_v[r + 1] + bv};
          const tc_f32x2 g = {acc_g[r] + bg, acc_g[r + 1] + bg};
          const tc_f32x2 h = (ABL & 1) ? v * g : v * gelu_erf_f2(g);
          char* const dst = hbase[(kr & 1) + (kr >> 2) * 2] + cr * 128;
          *reinterpret_cast<bf16_t*>(dst) = (bf16_t)h[0];
          *reinterpret_cast<bf16_t*>(dst + 128) = (bf16_t)h[1];
          if ((r + 2) % GI == 0) __builtin_amdgcn_sched_barrier(0);   // GI values in flight: the chain of a GELU is ~20 dependent instructions
        }
        ++cw2;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the hidden values are in LDS, the W2 slice has landed
        bar();
      }
      // ---- X: the other group's G -- its threads' pieces of the W2 slice are only known to have landed after ITS drain --
      // an empty interval for this group (the hidden rows a group reads are written by that group alone)
      bar();
      // ---- ff2: [32 x 64] hidden (A, from LDS) x W2 slice [160 x 64] (B, from LDS) -> out_acc, 20 MFMAs
      {
        const char* hb = smem + FF_H_OFF + (c & 1) * FF_H_BYTES + h_row();
        const char* wb = smem + FF_W2_OFF + w2_row();
        bf16x8 ha[4], b2[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ha[kk] = *reinterpret_cast<const bf16x8*>(hb + coff(kk));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) b2[j][kk] = *reinterpret_cast<const bf16x8*>(wb + j * 4096 + coff(kk));
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if constexpr (ABL & 8) asm volatile("" : "+v"(out_acc[j]) : "v"(ha[kk]), "v"(b2[j & 1][kk]));
            else out_acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[kk], b2[j & 1][kk], out_acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (j + 2 < 5) {                            // the set just consumed is refilled two blocks ahead
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b2[j & 1][kk] = *reinterpret_cast<const bf16x8*>(wb + (j + 2) * 4096 + coff(kk));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        bar();
      }
    }

    if (grp == 0) g8_barrier();                     // realign: every wave has executed the same number of barriers
    // ---- epilogue: + b2 + residual (the raw rows), bf16.  Per wave six passes of (16 rows x 64 columns) through a
    // private 4 KiB slab carved from the hidden-chunk buffers: wave (wm, wn) takes buffer wn, rows wm*32..+32 -- rows only
    // its own group reads, and this group's last read of them is behind it
    {
      float* slab = reinterpret_cast<float*>(smem + FF_H_OFF + wn * FF_H_BYTES + wm * 32 * 128);
      const int vc = lane & 7, lr0 = lane >> 3;
      auto pass = [&](auto J0_, auto NJ_, auto HALF_) {
        constexpr int j0 = decltype(J0_)::value, nj = decltype(NJ_)::value, half = decltype(HALF_)::value;
#pragma unroll
        for (int j = 0; j < nj; ++j)
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            const int r = 8 * half + qq;
            const int lr = (r & 3) + 4 * fhalf + 8 * ((r >> 2) & 1);
            slab[lr * 64 + j * 32 + frow] = out_acc[j0 + j][r];
          }
        const int n0 = wn * 160 + j0 * 32 + vc * 8;
        if (vc * 8 < nj * 32) {
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int lr = lr0 + 8 * qq;
            const int m = tile * FF_BM + wm * 32 + half * 16 + lr;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * 64 + vc * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * 64 + vc * 8 + 4);
            if (m < p.m) {
              float xv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              float rf[8];
              unpack8(*reinterpret_cast<const u32x4*>(p.x + (int64_t)m * p.ldx + n0), rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = (xv[e] + bl[2 * FF_H + n0 + e]) + rf[e];
              *reinterpret_cast<u32x4*>(p.out + (int64_t)m * p.ldo + n0) = pack8(xv);
            }
          }
        }
      };
      pass(ic<0>{}, ic<2>{}, ic<0>{});
      pass(ic<0>{}, ic<2>{}, ic<1>{});
      pass(ic<2>{}, ic<2>{}, ic<0>{});
      pass(ic<2>{}, ic<2>{}, ic<1>{});
      pass(ic<4>{}, ic<1>{}, ic<0>{});
      pass(ic<4>{}, ic<1>{}, ic<1>{});
    }
  }
  ff_wait_vmcnt<0>();                               // the stream ran ahead: nothing may land in LDS after the block is gone
}

int ff_mode() {        // TC_FF_FUSED = 0 never | 1 (default) whenever the shape is the level-0 block's; read per call
  const char* e = getenv("TC_FF_FUSED");
  return e ? atoi(e) : 1;
}

}  // namespace

extern "C" int tc_ff_geglu_fused_eligible(const TcFfParams* p) {
  if (!p || ff_mode() == 0) return 0;
  if (p->c != FF_C || p->hidden != FF_H || p->m <= 0) return 0;
  if (p->ldx < FF_C || p->ldo < FF_C || (p->ldx & 7) || (p->ldo & 7)) return 0;
  if ((int64_t)p->m * p->ldx * 2 >= 0x7fffffffLL * 64) return 0;
  return 1;
}

extern "C" int tc_ff_geglu_fused(const TcFfParams* p, void* stream) {
  if (!p || !p->x || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->out) return TC_EINVAL;
  if (!tc_ff_geglu_fused_eligible(p)) return TC_ESHAPE;
  if (!tc_aligned16(p->x) || !tc_aligned16(p->w1) || !tc_aligned16(p->w2) || !tc_aligned16(p->out)) return TC_EALIGN;
  FfArgs a;
  a.x = reinterpret_cast<const bf16_t*>(p->x); a.w1 = reinterpret_cast<const bf16_t*>(p->w1); a.b1 = p->b1;
  a.w2 = reinterpret_cast<const bf16_t*>(p->w2); a.b2 = p->b2; a.out = reinterpret_cast<bf16_t*>(p->out);
  a.m = p->m; a.ldx = p->ldx; a.ldo = p->ldo; a.ln = p->ln ? 1 : 0; a.eps = p->ln_eps;
  a.tiles = (p->m + FF_BM - 1) / FF_BM;
#ifdef TC_TIMING_BUILDS      /* interval trace / timing ablations: WRONG results by construction, never in the product library */
  a.trace = [&]() -> unsigned long long* { const char* e = getenv("TC_FF_TRACE"); return e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }();
#else
  a.trace = nullptr;
#endif
  static const int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int gmax = [&] { const char* e = getenv("TC_FF_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : cus; }();
  // every block the same number of tiles: 640 tiles on 256 CUs are three rounds either way, and 214 blocks of three
  // leave the weight stream (L2 -> LDS, shared by all) less contended than 256 blocks of two or three
  const int rounds = (a.tiles + gmax - 1) / gmax;
  const int grid = (a.tiles + rounds - 1) / rounds;
#ifdef TC_TIMING_BUILDS
  const int abl = [&] { const char* e = getenv("TC_FF_ABLATE"); return e ? atoi(e) : 0; }();
#endif
  const int gi = [&] { const char* e = getenv("TC_FF_GILP"); return e ? atoi(e) : 8; }();
  const int la = [&] { const char* e = getenv("TC_FF_LOOKAHEAD"); return e ? atoi(e) : 3; }();
  const dim3 g((unsigned)grid), b(FF_THREADS);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define FF_LAUNCH(LA_, ABL_) hipLaunchKernelGGL((ff_fused_kernel<LA_, ABL_, 2>), g, b, 0, st, a)
#ifdef TC_TIMING_BUILDS
  if (abl == 1) FF_LAUNCH(3, 1);
  else if (abl == 2) FF_LAUNCH(3, 2);
  else if (abl == 3) FF_LAUNCH(3, 3);
  else if (abl == 12) FF_LAUNCH(3, 12);
  else if (abl == 15) FF_LAUNCH(3, 15);
  else if (abl == 16 && a.trace) FF_LAUNCH(3, 16);
  else
#endif
  if (gi == 2) FF_LAUNCH(3, 0);
  else if (la == 2) FF_LAUNCH(2, 0);
  else if (la == 4) hipLaunchKernelGGL((ff_fused_kernel<4, 0, 8>), g, b, 0, st, a);
  else hipLaunchKernelGGL((ff_fused_kernel<3, 0, 8>), g, b, 0, st, a);       // the product kernel
#undef FF_LAUNCH
  TC_LAUNCH_CHECK();
  return TC_OK;
}
