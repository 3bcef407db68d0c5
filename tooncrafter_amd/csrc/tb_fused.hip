// Fused temporal self-attention of a level-0 transformer block, gfx950:
//
//     out = x + Wo . Attn_frames( Wqkv . LayerNorm(x) + bqkv ) + bo        x: [B*16*HW, 320] bf16, 5 heads of 64
//
// (reference lvdm/modules/attention.py:81-144 CrossAttention over the T = 16 frames of a pixel, called from
// TemporalTransformer, attention.py:365-412, behind norm1 / norm2 of BasicTransformerBlock, attention.py:225-246).  As four
// launches -- LayerNorm(-prologue) qkv projection (81920 x 960 x 320), tc_attn_temporal, output projection + residual --
// the block writes a 157 MB qkv tensor and a 52 MB attention output to HBM and reads both back for 0.018 TFLOP of
// attention arithmetic.  Here neither exists.  The skeleton is csrc/ff_fused.hip's:
//
//  * a block owns 8 consecutive pixels x 16 frames = 128 GATHERED rows (tile row = pixel * 16 + frame; a pixel's frames are
//    HW rows apart in memory, each row 640 contiguous bytes); wave (wm, wn) of its 8 waves owns rows wm*32..+32 = two pixels.
//    The rows' LayerNorm runs in registers on the MFMA A-operand layout and stays there for the whole tile;
//  * the five heads are walked one after the other.  Per head two "stages" of five K-steps against [128 rows x 64 k] tiles
//    of Wqkv: stage A = the head's 64 q rows | its 64 k rows (waves wn = 0 produce q, wn = 1 produce k, 8 MFMAs per K-step),
//    stage B = its 64 v rows (waves wn = 0 only).  q and k go to LDS row-major, v TRANSPOSED ([64 dims][128 rows], the B
//    operand of P.V); then every wave runs the attention of ONE pixel (wave (wm, wn): pixel 2 wm + wn) on 16x16x32 MFMAs:
//    S^T = K Q^T (a lane owns one query: softmax reductions in-lane + two shuffles), P re-laid as the A operand by four
//    ds_bpermute, O = P V, written as bf16 in A layout over the head's q rows; then 20 MFMAs of the output projection
//    against Wo's [320 x 64] slice into the wave's 32 x 160 output accumulators, which live across the heads;
//  * weights (819 KB, L2-resident) stream by LDS-DMA from inline asm: Wqkv K-tiles through a ring of three 16 KiB stages
//    (requested two steps ahead; the stream runs on across heads and tiles), Wo's slice once per head in five pieces;
//    every wait is a hand-counted vmcnt;
//  * the two wave groups (wm >> 1: one wave per SIMD each) run one barrier interval apart, as in ff_fused.hip / gemm8.hip.
//
// LDS: W ring 48 KiB | Wo slice 40 | q (then the head's output) 16 | k 16 | v^T 17 | biases 5 | parked A fragments 12 = 154 KiB.
// Roundings: LayerNorm output, q / k / v, the softmax weights and the attention output in bf16, sums in fp32 -- the
// roundings of the four launches (tc_attn_temporal keeps its softmax weights in fp32: the one difference).
#include "gemm_common.h"
#include "gemm_persist.h"

#include <stdlib.h>

namespace {

constexpr int TB_C = 320, TB_HEADS = 5, TB_T = 16, TB_BM = 128, TB_THREADS = 512;
constexpr int TB_KT = TB_C / TC_BK;               // 5 K-steps per stage
constexpr int TB_W_STAGE = 128 * 128;             // 16 KiB: 128 rows x 64 k
constexpr int TB_NRING = 3;
constexpr int TB_W_OFF = 0;
constexpr int TB_WO_OFF = TB_NRING * TB_W_STAGE;  // 40 KiB: 320 rows x 64 k
constexpr int TB_WO_BYTES = 320 * 128;
constexpr int TB_Q_OFF = TB_WO_OFF + TB_WO_BYTES; // [128 rows][64] bf16, 16-byte chunks XOR-swizzled by (row >> 1) & 7
constexpr int TB_K_OFF = TB_Q_OFF + 128 * 128;
constexpr int TB_VT_OFF = TB_K_OFF + 128 * 128;   // [64 dims][128 rows + 8] bf16: 272-byte rows (conflict-free 16-lane reads)
constexpr int TB_VT_LD = 272;
constexpr int TB_B_OFF = TB_VT_OFF + 64 * TB_VT_LD;          // bqkv (960 fp32) | bo (320 fp32)
constexpr int TB_P_OFF = TB_B_OFF + (3 * TB_C + TB_C) * 4;   // parked A fragments: [slot][wm][lane] x 16 B
constexpr int TB_NPARK = 3;
constexpr int TB_NRES = 20 - TB_NPARK;
constexpr int TB_LDS = TB_P_OFF + TB_NPARK * 4096;
static_assert(TB_LDS <= 160 * 1024, "LDS");
static_assert(TB_VT_OFF + 64 * TB_VT_LD - TB_K_OFF >= 8 * 4096, "epilogue slabs live in the k / v^T buffers");

struct TbArgs {
  const bf16_t* x; const bf16_t* wqkv; const float* bqkv; const bf16_t* wo; const float* bo; bf16_t* out;
  int hw, ldx, ldo, ln;
  float eps, scale_log2e;
  int tiles, tiles_per_b;
  int abl;          // timing ablations (TC_TB_ABLATE; wrong results): 1 no head loop (row loads, LayerNorm, epilogue only), 2 no row loads, 4 no epilogue, 8 interval trace (TC_TB_TRACE)
  unsigned long long* trace;   // TC_TB_TRACE (with TC_TB_ABLATE bit 8): s_memtime after every barrier of block 0's waves 0 and 4
  int stagger;      // TC_TB_STAGGER: block i starts (i & 3) * stagger * ~3.4 us late (de-phases the blocks' memory phases)
};

// v_permlane16_swap / v_permlane32_swap with both operands = x: every lane gets (a, b) = the values its 16-lane row pair
// (lanes l and l ^ 16) / its half pair (l and l ^ 32) hold, lower row's first -- a cross-row exchange on the VALU instead of a
// ds_bpermute round trip through LDS
__device__ __forceinline__ void tb_swap16(uint32_t x, uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void tb_swap32(uint32_t x, uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ float tb_max_rows(float x) {         // max over the four lanes l15 + 16 g
  uint32_t a, b;
  tb_swap16(__builtin_bit_cast(uint32_t, x), a, b);
  x = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  tb_swap32(__builtin_bit_cast(uint32_t, x), a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float tb_sum_rows(float x) {
  uint32_t a, b;
  tb_swap16(__builtin_bit_cast(uint32_t, x), a, b);
  x = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  tb_swap32(__builtin_bit_cast(uint32_t, x), a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

template <int N>
__device__ __forceinline__ void tb_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool TRACE>
__global__ __launch_bounds__(TB_THREADS, 2) void tb_fused_kernel(const TbArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[TB_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int grp = wave_u >> 2;                   // = wm >> 1: waves w and w + 4 share a SIMD, one of each group
  const int frow = lane & 31, fhalf = lane >> 5;

  const g8_srd_t w_srd = g8_make_srd(p.wqkv, (int64_t)3 * TB_C * TB_C * 2);
  const g8_srd_t wo_srd = g8_make_srd(p.wo, (int64_t)TB_C * TB_C * 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // biases into LDS once (no global load may sit inside the head loop: hipcc would wait vmcnt(0) for it and drain the stream)
  {
    float* bl = reinterpret_cast<float*>(smem + TB_B_OFF);
    for (int i = tid; i < 3 * TB_C; i += TB_THREADS) bl[i] = p.bqkv[i];
    for (int i = tid; i < TB_C; i += TB_THREADS) bl[3 * TB_C + i] = p.bo[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  // ---- weight stream: thread -> (row lrow of a 64-row piece, 16-byte chunk); the XOR swizzle on the SOURCE chunk.  Wqkv
  // and Wo both have 320-element rows, so one per-lane offset serves both
  const int lrow = tid >> 3;
  const int sch = (tid & 7) ^ ((lrow >> 1) & 7);
  const uint32_t voff = (uint32_t)(lrow * TB_C * 2 + sch * 16);
  const uint32_t dma_dst = lds0 + wave_u * 1024;
  // K-tile q of the cyclic stream (50 per tile: head h = q / 10, stage (q % 10) / 5, K-step q % 5) -> ring stage q % 3.
  // Stage A: rows 0..63 <- the head's q rows (h*64 ..), rows 64..127 <- its k rows (320 + h*64 ..): two pieces per thread;
  // stage B: rows 0..63 <- its v rows (640 + h*64 ..): one piece
  auto dma_w = [&](int q) {
    const int qq = q % (TB_HEADS * 2 * TB_KT);
    const int h = qq / (2 * TB_KT), r = qq - h * 2 * TB_KT;
    const int kt = r < TB_KT ? r : r - TB_KT;
    const uint32_t dst = dma_dst + TB_W_OFF + (q % TB_NRING) * TB_W_STAGE;
    if (r < TB_KT) {
      const uint32_t so = (uint32_t)((h * 64 * TB_C + kt * TC_BK) * 2);
      g8_dma16(w_srd, dst, voff, so);
      g8_dma16(w_srd, dst + 8192, voff, so + TB_C * TB_C * 2);
    } else {
      const uint32_t so = (uint32_t)(((2 * TB_C + h * 64) * TB_C + kt * TC_BK) * 2);
      g8_dma16(w_srd, dst, voff, so);
    }
  };
  auto dma_wo = [&](int h, int piece) {             // rows 64 piece .. +64 of Wo's slice for head h (columns h*64 .. +64)
    const uint32_t so = (uint32_t)((piece * 64 * TB_C + h * 64) * 2);
    g8_dma16(wo_srd, dma_dst + TB_WO_OFF + piece * 8192, voff, so);
  };

  // ---- fragment addressing (32x32x16 MFMA): lane holds row lane & 31, k = 8 (lane >> 5) .. of slice kk -> chunk 2 kk + (lane >> 5)
  auto coff = [&](int kk) { return ((kk * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4; };

  bf16x8 xa[TB_NRES];                               // the tile's normalised rows: K-slices 0 .. TB_NRES-1 (the rest parked in LDS)
  char* const park = smem + TB_P_OFF + wm * 1024 + lane * 16;
  f32x16 out_acc[5];
  f32x16 acc_v, acc_g;                              // the stage's 32 x 64 block of the wave: columns 0..31 | 32..63

  // TRACE build (abl bit 8): per-interval timing.  Block 0, waves 0 (group 0) and 4 (group 1), second tile, first two heads: the shader clock
  // after every barrier -> trace[wave >> 2][n] (a store per barrier: the timed build is not the product kernel's schedule to the
  // cycle, its waits are the same)
  int tr_n = 0;
  bool tr_on = false;
  auto bar = [&]() {
    g8_barrier();
    if constexpr (TRACE) {
      if (tr_on && tr_n < 64) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) p.trace[(wave_u >> 2) * 64 + tr_n] = t;
        ++tr_n;
      }
    }
  };
  for (int i = (blockIdx.x & 3) * p.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(127);
  int q = 0;                                        // K-tile stream position consumed next
  dma_w(0);
  dma_w(1);
  tb_wait_vmcnt<0>();
  g8_barrier();

  const float* bl = reinterpret_cast<const float*>(smem + TB_B_OFF);

  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    const int bb = tile / p.tiles_per_b;
    const int p0 = (tile - bb * p.tiles_per_b) * 8;
    // tile row lr = pixel * 16 + frame -> memory row (bb * 16 + frame) * hw + p0 + pixel
    auto grow = [&](int lr) { return (int64_t)(bb * TB_T + (lr & 15)) * p.hw + p0 + (lr >> 4); };
    {
      u32x4 raw[20];
      const bf16_t* xr = p.x + grow(wm * 32 + frow) * p.ldx + 8 * fhalf;
#pragma unroll
      for (int s = 0; s < 20; ++s) raw[s] = (p.abl & 2) ? u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : *reinterpret_cast<const u32x4*>(xr + 16 * s);
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
        const float mean = sum * (1.0f / TB_C);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < 20; ++s) {
          float f[8];
          unpack8(raw[s], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq += d * d; }
        }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / TB_C) + p.eps);
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
        if (s < TB_NRES) xa[s] = __builtin_bit_cast(bf16x8, raw[s]);
        else *reinterpret_cast<u32x4*>(park + (s - TB_NRES) * 4096) = raw[s];
      }
#pragma unroll
      for (int s = 0; s < TB_NRES; ++s) asm volatile("" ::"v"(xa[s]));     // row loads consumed before the head loop (ff_fused.hip)
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out_acc[j][r] = 0.f;

    if (grp == 1) g8_barrier();                     // the stagger, per tile (ff_fused.hip)
    for (int h = (p.abl & 1) ? TB_HEADS : 0; h < TB_HEADS; ++h) {
      if constexpr (TRACE) tr_on = blockIdx.x == 0 && (wave_u & 3) == 0 && tile == (int)(blockIdx.x + gridDim.x) && h < 2;
      // ---- one K-step of a stage: fragments of W K-tile q (ring stage q % 3), K-tile q + 2 and this step's share of Wo's
      // slice requested, 8 MFMAs.  Requests per head in program order (pieces per thread):
      //   A0: W 2 | A1: W 2 | A2: W 2 | A3: W 1 | A4: W 1 | B0: W 1 | B1: W 1, p0, p1 | B2: W 1, p2 | B3: W 2, p3 | B4: W 2, p4
      // (the tile requested at step pos is pos + 2: a stage-A tile has two pieces, a stage-B tile one); everything is
      // drained at the end of the v write-out.  The wait at the end of a step's read segment retires K-tile q + 1 (requested
      // one step earlier, first in that step's requests); what may stay in flight is what was requested after it:
      //   A0 2 | A1 2 | A2 2 | A3 1 | A4 1 | B0 1 | B1 3 | B2 4 | B3 4 | B4 4
      auto step = [&](auto STAGE_, auto S_) {
        constexpr int stage = decltype(STAGE_)::value, s = decltype(S_)::value;
        const bool act = stage == 0 || wn == 0;       // stage B: the k-side waves have no columns
        const char* st = smem + TB_W_OFF + (q % TB_NRING) * TB_W_STAGE + (wn * 64 + frow) * 128;
        bf16x8 bw[2][4];
        if (act) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bw[j][kk] = *reinterpret_cast<const bf16x8*>(st + j * 4096 + coff(kk));
        }
        dma_w(q + 2);
        if (stage == 1 && s == 1) { dma_wo(h, 0); dma_wo(h, 1); }
        if (stage == 1 && s >= 2) dma_wo(h, s);
        constexpr int pos = stage * 5 + s;
        constexpr int keep = pos <= 2 ? 2 : (pos <= 5 ? 1 : (pos == 6 ? 3 : 4));
        tb_wait_vmcnt<keep>();
        bar();
        __builtin_amdgcn_s_setprio(1);
        if (act) {
          auto mm = [&](auto KK_) {
            constexpr int kk = decltype(KK_)::value, ks = 4 * s + kk;
            if constexpr (ks < TB_NRES) {
              acc_v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks], bw[0][kk], acc_v, 0, 0, 0);
              acc_g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks], bw[1][kk], acc_g, 0, 0, 0);
            } else {
              const bf16x8 pa = *reinterpret_cast<const bf16x8*>(park + (ks - TB_NRES) * 4096);
              acc_v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, bw[0][kk], acc_v, 0, 0, 0);
              acc_g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, bw[1][kk], acc_g, 0, 0, 0);
            }
            if constexpr (s == 4) __builtin_amdgcn_sched_barrier(0);
          };
          mm(ic<0>{});
          mm(ic<1>{});
          mm(ic<2>{});
          mm(ic<3>{});
        }
        __builtin_amdgcn_s_setprio(0);
        bar();
        ++q;
      };
      // ---- write-out of a stage's block: + bias, bf16.  q / k: row-major [128][64], 16-byte chunks XOR-swizzled by
      // (row >> 1) & 7 (accumulator register r of a lane = row cr = (r & 3) + 8 (r >> 2) + 4 fhalf of the wave's 32, column
      // frow | 32 + frow: (row >> 1) & 7 = kr | 2 fhalf with kr in {0, 1, 4, 5}, so four lane-dependent bases serve all 32
      // stores -- ff_fused.hip); v: transposed, [64 dims][128 rows], four consecutive rows of a lane as one 8-byte store
      auto lane_now = [&]() {                          // the lane id afresh and opaque: nothing hoisted out of the head loop
        int gl = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(gl));
        return gl;
      };
      auto write_qk = [&]() {
        const int gl = lane_now();
        const int fr = gl & 31, fh = gl >> 5;
        const float b0 = bl[wn * TB_C + h * 64 + fr], b1 = bl[wn * TB_C + h * 64 + 32 + fr];
        char* const hrow = smem + (wn ? TB_K_OFF : TB_Q_OFF) + (wm * 32 + 4 * fh) * 128 + (fr & 7) * 2;
        const int a2 = (fr >> 3) ^ (2 * fh);
        char* const hb[4] = {hrow + (a2 << 4), hrow + ((a2 ^ 1) << 4), hrow + ((a2 ^ 4) << 4), hrow + ((a2 ^ 5) << 4)};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const int kr = (cr >> 1) & 7;                // 0, 1, 4 or 5
          const int i0 = (kr & 1) + (kr >> 2) * 2;      // base of column frow; column 32 + frow: chunk ^ 4 -> kr ^ 4
          *reinterpret_cast<bf16_t*>(hb[i0] + cr * 128) = (bf16_t)(acc_v[r] + b0);
          *reinterpret_cast<bf16_t*>(hb[i0 ^ 2] + cr * 128) = (bf16_t)(acc_g[r] + b1);
        }
      };
      auto write_vt = [&]() {
        if (wn != 0) return;
        const int gl = lane_now();
        const int fr = gl & 31, fh = gl >> 5;
        const float b0 = bl[2 * TB_C + h * 64 + fr], b1 = bl[2 * TB_C + h * 64 + 32 + fr];
        char* const v0 = smem + TB_VT_OFF + fr * TB_VT_LD + (wm * 32 + 4 * fh) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // rows 8 i + 4 fhalf + (0..3) of the wave's 32
          uint32_t lo[2], hi[2];
          lo[0] = pack2(acc_v[4 * i] + b0, acc_v[4 * i + 1] + b0);
          lo[1] = pack2(acc_v[4 * i + 2] + b0, acc_v[4 * i + 3] + b0);
          hi[0] = pack2(acc_g[4 * i] + b1, acc_g[4 * i + 1] + b1);
          hi[1] = pack2(acc_g[4 * i + 2] + b1, acc_g[4 * i + 3] + b1);
          *reinterpret_cast<uint2*>(v0 + i * 16) = uint2{lo[0], lo[1]};
          *reinterpret_cast<uint2*>(v0 + 32 * TB_VT_LD + i * 16) = uint2{hi[0], hi[1]};
        }
      };

#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_v[r] = 0.f; acc_g[r] = 0.f; }
      step(ic<0>{}, ic<0>{});
      step(ic<0>{}, ic<1>{});
      step(ic<0>{}, ic<2>{});
      step(ic<0>{}, ic<3>{});
      step(ic<0>{}, ic<4>{});
      write_qk();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_v[r] = 0.f; acc_g[r] = 0.f; }
      step(ic<1>{}, ic<0>{});
      step(ic<1>{}, ic<1>{});
      step(ic<1>{}, ic<2>{});
      step(ic<1>{}, ic<3>{});
      step(ic<1>{}, ic<4>{});
      write_vt();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // q, k, v of the head are in LDS; this thread's Wo pieces landed
      bar();

      // ---- attention of ONE pixel per wave (rows pr .. pr + 16 of the tile = its 16 frames), 16x16x32 MFMAs.
      // (The other group's Wo pieces are only known to have landed after ITS drain, one interval behind this one: this
      // interval separates that drain from the output projection's reads, as the empty interval of ff_fused.hip does.)
      {
        const int gl = lane_now();
        const int l15 = gl & 15, g4 = gl >> 4;
        const int pr = wm * 32 + wn * 16;
        const int row = pr + l15;
        const int sw = (row >> 1) & 7;
        const char* qrow = smem + TB_Q_OFF + row * 128;
        const char* krow = smem + TB_K_OFF + row * 128;
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {               // S^T[key][query] = sum_d K[key][d] Q[query][d]
          const int c = ((ks * 4 + g4) ^ sw) << 4;
          const bf16x8 ka = *reinterpret_cast<const bf16x8*>(krow + c);
          const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qrow + c);
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qb, st, 0, 0, 0);
        }
        // lane: query l15, keys 4 g4 + r.  Softmax over the 16 keys: in-lane over r, across g4 by two shuffles
        float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
        mx = tb_max_rows(mx);
        float e[4], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f((st[r] - mx) * p.scale_log2e); sum += e[r]; }
        sum = tb_sum_rows(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        const uint32_t pk0 = pack2(e[0] * inv, e[1] * inv), pk1 = pack2(e[2] * inv, e[3] * inv);
        // P as the A operand of P.V (rows = queries, k = keys 8 g' .. +7, keys 16..31 of the 32-deep slice are zero):
        // lane (query, g' = 0) <- keys 0..3 (own) | 4..7 (lane + 16); (query, 1) <- 8..11 (lane + 16) | 12..15 (lane + 32)
        // (rows of 16 lanes r0..r3 = g4: swap16(x) gives row 0 (x.r0, x.r1); swap32(x)'s second value brings rows 2, 3 down
        // to rows 0, 1, and swap16 of THAT gives row 1 (x.r2, x.r3))
        uint32_t a0, b0, a1, b1, lo, hi, c0, d0, c1, d1;
        tb_swap16(pk0, a0, b0);
        tb_swap16(pk1, a1, b1);
        tb_swap32(pk0, lo, hi);
        tb_swap16(hi, c0, d0);
        tb_swap32(pk1, lo, hi);
        tb_swap16(hi, c1, d1);
        const bool r0 = g4 == 0, r1 = g4 == 1;
        u32x4 pw;
        pw[0] = r0 ? a0 : (r1 ? c0 : 0u);
        pw[1] = r0 ? a1 : (r1 ? c1 : 0u);
        pw[2] = r0 ? b0 : (r1 ? d0 : 0u);
        pw[3] = r0 ? b1 : (r1 ? d1 : 0u);
        const bf16x8 pa = __builtin_bit_cast(bf16x8, pw);
        // O[query][d] = sum_key P[query][key] V[key][d]: B operand from v^T (lane: dim db*16 + l15, keys 8 g' .. +7 of the
        // pixel; g' >= 2 meets the zero half of P: it re-reads the valid half, never uninitialised bytes)
        const char* vt = smem + TB_VT_OFF + l15 * TB_VT_LD + (pr + 8 * (g4 & 1)) * 2;
        f32x4_t od[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 vb = *reinterpret_cast<const bf16x8*>(vt + db * 16 * TB_VT_LD);
          od[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // lane: dim db*16 + l15, queries 4 g4 + r -> the head's output over its q rows, A layout of the projection
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int orow = pr + 4 * g4 + r;
            const int d = db * 16 + l15;
            char* dst = smem + TB_Q_OFF + orow * 128 + (((d >> 3) ^ ((orow >> 1) & 7)) << 4) + (d & 7) * 2;
            *reinterpret_cast<bf16_t*>(dst) = (bf16_t)od[db][r];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
      }

      // ---- output projection: [32 x 64] head output (A, from LDS) x Wo slice [160 x 64] (B, from LDS) -> out_acc, 20 MFMAs
      {
        const char* hb = smem + TB_Q_OFF + (wm * 32 + frow) * 128;
        const char* wb = smem + TB_WO_OFF + (wn * 160 + frow) * 128;
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
            out_acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[kk], b2[j & 1][kk], out_acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (j + 2 < 5) {
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
    // ---- epilogue: + bo + residual (the raw rows), bf16, through a private 4 KiB slab per wave carved from the k / v^T
    // buffers (dead: the last head's attention is behind every wave)
    {
      float* slab = reinterpret_cast<float*>(smem + TB_K_OFF + wave_u * 4096);
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
            const int64_t m = grow(wm * 32 + half * 16 + lr);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * 64 + vc * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * 64 + vc * 8 + 4);
            float xv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            float rf[8];
            unpack8(*reinterpret_cast<const u32x4*>(p.x + m * p.ldx + n0), rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = (xv[e] + bl[3 * TB_C + n0 + e]) + rf[e];
            *reinterpret_cast<u32x4*>(p.out + m * p.ldo + n0) = pack8(xv);
          }
        }
      };
      if (!(p.abl & 4)) {
        pass(ic<0>{}, ic<2>{}, ic<0>{});
        pass(ic<0>{}, ic<2>{}, ic<1>{});
        pass(ic<2>{}, ic<2>{}, ic<0>{});
        pass(ic<2>{}, ic<2>{}, ic<1>{});
        pass(ic<4>{}, ic<1>{}, ic<0>{});
        pass(ic<4>{}, ic<1>{}, ic<1>{});
      }
    }
  }
  tb_wait_vmcnt<0>();                               // the stream ran ahead: nothing may land in LDS after the block is gone
}

// TC_TB_FUSED = 1 whenever the shape is the level-0 block's | 0 never (the DEFAULT since round 6); read per call.
// Rounds 4-5 shipped this kernel as the level-0 route: 1.39x the four launches it replaced (LayerNorm, qkv projection,
// tc_attn_temporal, output projection).  Round 6's csrc/qkv_attn.hip made a better THREE-launch chain of those -- LayerNorm,
// projection + attentions in one launch (72 us at this width where the two launches took 145), the weight-stationary
// output projection -- and in the same-process A/B of the guided forward that chain is ahead of this kernel on both leases
// tried: +0.57 % and +0.38 % (profiles/r06_l0_chain_vs_tb_fused_forward_ab*.txt).  This kernel is LDS-bandwidth-bound
// (header); the chain's launches are HBM-bound and each runs near its own roof.  Kept, tested, one switch away.
int tb_mode() {
  const char* e = getenv("TC_TB_FUSED");
  return e ? atoi(e) : 0;
}

}  // namespace

extern "C" int tc_temporal_attn_fused_eligible(const TcTbParams* p) {
  if (!p || tb_mode() == 0) return 0;
  if (p->c != TB_C || p->heads != TB_HEADS || p->t != TB_T || p->b <= 0 || p->hw <= 0 || (p->hw & 7)) return 0;
  if (p->ldx < TB_C || p->ldo < TB_C || (p->ldx & 7) || (p->ldo & 7)) return 0;
  if ((int64_t)p->b * p->t * p->hw * (p->ldx > p->ldo ? p->ldx : p->ldo) * 2 >= 0x7fffffffLL * 64) return 0;
  return 1;
}

extern "C" int tc_temporal_attn_fused(const TcTbParams* p, void* stream) {
  if (!p || !p->x || !p->wqkv || !p->bqkv || !p->wo || !p->bo || !p->out) return TC_EINVAL;
  if (!tc_temporal_attn_fused_eligible(p)) return TC_ESHAPE;
  if (!tc_aligned16(p->x) || !tc_aligned16(p->wqkv) || !tc_aligned16(p->wo) || !tc_aligned16(p->out)) return TC_EALIGN;
  TbArgs a;
  a.x = reinterpret_cast<const bf16_t*>(p->x); a.wqkv = reinterpret_cast<const bf16_t*>(p->wqkv); a.bqkv = p->bqkv;
  a.wo = reinterpret_cast<const bf16_t*>(p->wo); a.bo = p->bo; a.out = reinterpret_cast<bf16_t*>(p->out);
  a.hw = p->hw; a.ldx = p->ldx; a.ldo = p->ldo; a.ln = p->ln ? 1 : 0; a.eps = p->ln_eps;
  a.scale_log2e = p->scale * 1.44269504088896340736f;
  a.tiles_per_b = p->hw / 8;
  a.tiles = p->b * a.tiles_per_b;
  static const int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
#ifdef TC_TIMING_BUILDS      /* timing ablations / interval trace: WRONG results by construction, never in the product library */
  a.abl = [&] { const char* e = getenv("TC_TB_ABLATE"); return e ? atoi(e) : 0; }();
  a.trace = [&]() -> unsigned long long* { const char* e = getenv("TC_TB_TRACE"); return e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }();
  if (!a.trace) a.abl &= ~8;
#else
  a.abl = 0;
  a.trace = nullptr;
#endif
  a.stagger = [&] { const char* e = getenv("TC_TB_STAGGER"); return e ? atoi(e) : 0; }();
  const int gmax = [&] { const char* e = getenv("TC_TB_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : cus; }();
  const int rounds = (a.tiles + gmax - 1) / gmax;
  const int grid = (a.tiles + rounds - 1) / rounds;
#ifdef TC_TIMING_BUILDS
  if (a.abl & 8) hipLaunchKernelGGL(tb_fused_kernel<true>, dim3((unsigned)grid), dim3(TB_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  else
#endif
  hipLaunchKernelGGL(tb_fused_kernel<false>, dim3((unsigned)grid), dim3(TB_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  TC_LAUNCH_CHECK();
  return TC_OK;
}
