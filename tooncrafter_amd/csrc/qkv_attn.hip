// Temporal self-attention behind its qkv projection as ONE launch, gfx950 (ABI 13):
//
//     out[:, h*64 .. h*64+64] = Attn_frames( x . Wqkv[q_h | k_h | v_h]^T + bqkv )          x: [B*16*HW, C] bf16
//
// (reference lvdm/modules/attention.py:81-144 CrossAttention over the T = 16 frames of a pixel -- to_q / to_k / to_v,
// 16 x 16 softmax per head, heads re-concatenated in front of to_out -- called from TemporalTransformer,
// attention.py:365-412, behind norm1 / norm2 of BasicTransformerBlock, attention.py:225-246.)  At UNet levels 1 and 2
// (C = 640 / 1280) the two launches this replaces are tc_gemm_bf16 (20480 x 1920 x 640, 5120 x 3840 x 1280) and
// tc_attn_temporal: the projection's epilogue is bound by its 79 / 39 MB of stores (chip-wide store ceiling ~3 TB/s,
// DESIGN 5.7) and the attention kernel reads those bytes straight back for 0.01 TFLOP of arithmetic.  Here the qkv
// tensor never exists: a third of the bytes are written, one launch instead of two.  (Level 0 has the larger fusion,
// csrc/tb_fused.hip, which also swallows the LayerNorm and the output projection: its rows are 320 wide and fit a wave's
// registers; at 640 / 1280 columns they do not, and the projection weights -- 2.4 / 9.8 MB per block tile -- would be
// re-streamed per 128 rows at the L2 -> LDS line.  DESIGN 5.9 prices that; this kernel is the part of it that pays.)
//
//  * a block owns 8 consecutive pixels x 16 frames = 128 GATHERED rows (tile row = pixel * 16 + frame; a pixel's frames are
//    HW rows apart in memory) and ONE head: a 128 x 192 output tile [q_h | k_h | v_h] over K = C, on the 4-wave skeleton
//    of csrc/gemm.hip -- tiles global -> LDS by buffer_load ... lds (XOR swizzle on the source side), two K-steps in
//    flight with counted vmcnt + raw s_barrier, waves 2 x 2, each 64 x 96 = 2 x 3 v_mfma_f32_32x32x16_bf16 sub-tiles;
//  * epilogue: + bias, bf16 (the roundings of the projection's own output), q and k row-major, v TRANSPOSED into the stage
//    memory; then every wave runs the attention of two pixels on 16x16x32 MFMAs exactly as tb_fused.hip does (S^T = K Q^T:
//    a lane owns one query, softmax in-lane + two cross-row swaps, P re-laid as the A operand by permlane swaps, O = P V),
//    writes O as bf16 over its pixels' q rows and stores those rows itself: 128 contiguous bytes per row, no block barrier
//    after the attention;
//  * blocks are dealt so that the heads of one row tile run back to back on one XCD (its L2 serves the A re-reads).
//
// LDS: 2 stages x (128 + 192) rows x 128 B = 80 KiB (two blocks per CU); the epilogue's q | k | v^T (16 + 16 + 17 KiB)
// live in the stage memory.  Roundings: q / k / v, the softmax weights and the attention output in bf16, sums in fp32 --
// tc_gemm_bf16 + tc_attn_temporal differ in ONE place (tc_attn_temporal keeps its softmax weights in fp32), as
// tb_fused.hip does.
#include "gemm_common.h"

#include <stdlib.h>

namespace {

constexpr int QA_BM = 128, QA_BN = 192, QA_THREADS = 256, QA_T = 16;
constexpr int QA_A_BYTES = QA_BM * TC_BK * 2;                 // 16 KiB
constexpr int QA_STAGE = (QA_BM + QA_BN) * TC_BK * 2;         // 40 KiB
constexpr int QA_LDS = 2 * QA_STAGE;                          // 80 KiB
constexpr int QA_Q_OFF = 0;                                   // [128 rows][64] bf16, 16-byte chunks XOR-swizzled by (row >> 1) & 7
constexpr int QA_K_OFF = 128 * 128;
constexpr int QA_VT_OFF = 2 * 128 * 128;                      // [64 dims][128 rows + 8] bf16: 272-byte rows
constexpr int QA_VT_LD = 272;
static_assert(QA_VT_OFF + 64 * QA_VT_LD <= QA_LDS, "epilogue buffers live in the stage memory");
constexpr int QA_RA = QA_BM / 32, QA_RB = QA_BN / 32;         // loader rows per thread: 4 + 6 requests per K-step

struct QaArgs {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* out;
  int hw, c, heads, ldx, ldo;
  float scale_log2e;
  int tiles, tiles_per_b;
};

template <int N>
__device__ __forceinline__ void qa_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// cross-row exchanges on the VALU (tb_fused.hip): every lane gets the values its 16-lane row pair / its half pair hold
__device__ __forceinline__ void qa_swap16(uint32_t x, uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void qa_swap32(uint32_t x, uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ float qa_max_rows(float x) {          // max over the four lanes l15 + 16 g
  uint32_t a, b;
  qa_swap16(__builtin_bit_cast(uint32_t, x), a, b);
  x = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  qa_swap32(__builtin_bit_cast(uint32_t, x), a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float qa_sum_rows(float x) {
  uint32_t a, b;
  qa_swap16(__builtin_bit_cast(uint32_t, x), a, b);
  x = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  qa_swap32(__builtin_bit_cast(uint32_t, x), a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

__global__ __launch_bounds__(QA_THREADS, 2) void qkv_attn_kernel(const QaArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[QA_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int frow = lane & 31, fhalf = lane >> 5;

  // block -> (row tile, head): XCD x (= blockIdx & 7) walks the row tiles x, x + 8, ..., all heads of a tile back to back
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / p.heads) * 8 + xcd;
  const int h = slot - (slot / p.heads) * p.heads;
  if (tile >= p.tiles) return;
  const int bb = tile / p.tiles_per_b;
  const int p0 = (tile - bb * p.tiles_per_b) * 8;
  // tile row lr = pixel * 16 + frame -> memory row row0 + frame * hw + pixel
  const int64_t row0 = (int64_t)bb * QA_T * p.hw + p0;

  // ---- loader geometry: thread -> (row lrow + 32 i, 16-byte chunk) of both tiles; the swizzle is on the SOURCE chunk
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const tc_rsrc_t a_rsrc = make_rsrc(p.x + row0 * p.ldx, (((int64_t)(QA_T - 1) * p.hw + 7) * p.ldx + p.c) * 2);
  const tc_rsrc_t w_rsrc = make_rsrc(p.w, (int64_t)3 * p.c * p.c * 2);
  uint32_t a_voff[QA_RA], b_voff[QA_RB];
#pragma unroll
  for (int i = 0; i < QA_RA; ++i) {
    const int lr = lrow + 32 * i;
    a_voff[i] = (uint32_t)((((int64_t)(lr & 15) * p.hw + (lr >> 4)) * p.ldx) * 2 + chunk * 16);
  }
#pragma unroll
  for (int i = 0; i < QA_RB; ++i) {
    // stage rows 0..63 <- to_q rows of head h, 64..127 <- to_k, 128..191 <- to_v (Wqkv = [q | k | v] blocks of C rows)
    const int r = lrow + 32 * i;
    b_voff[i] = (uint32_t)(((int64_t)((i >> 1) * p.c + h * 64 + (r & 63)) * p.c) * 2 + chunk * 16);
  }
  auto load_tile = [&](int kb, int stage) {
    const uint32_t soff = (uint32_t)kb * (TC_BK * 2);
    char* sa = smem + stage * QA_STAGE + wave_u * 1024;
    char* sb = sa + QA_A_BYTES;
#pragma unroll
    for (int i = 0; i < QA_RB; ++i) glds16(w_rsrc, sb + i * 4096, b_voff[i], soff);
#pragma unroll
    for (int i = 0; i < QA_RA; ++i) glds16(a_rsrc, sa + i * 4096, a_voff[i], soff);
  };

  // bias of this lane's column in each of the wave's three 32-column blocks (the projections of the reference have none:
  // bias == nullptr; a LayerNorm folded into Wqkv brings one)
  float bcol[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int jb = wn * 3 + j;                       // 32-column block of the 192: 0, 1 = q | 2, 3 = k | 4, 5 = v
    bcol[j] = p.bias ? p.bias[(jb >> 1) * p.c + h * 64 + (jb & 1) * 32 + frow] : 0.f;
  }

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const char* sa = smem + stage * QA_STAGE;
    const char* sb = sa + QA_A_BYTES;
    bf16x8 af[2][2], bf[2][3];
    auto frags = [&](int kk, bf16x8 (&a)[2], bf16x8 (&b)[3]) {
      const int c = kk * 2 + fhalf;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * 64 + i * 32 + frow, c));
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(wn * 96 + j * 32 + frow, c));
    };
    auto mfmas = [&](bf16x8 (&a)[2], bf16x8 (&b)[3]) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
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
  };

  // ---- K loop (csrc/gemm.hip, PIPE): both stages requested before the first wait, a stage re-requested as soon as every
  // wave has its fragments in registers; counted vmcnt + raw s_barrier (a __syncthreads() would drain the DMA queue)
  const int nk = p.c / TC_BK;
  load_tile(0, 0);
  if (nk > 1) load_tile(1, 1);
  for (int kb = 0; kb < nk; ++kb) {
    const int st = kb & 1;
    if (kb + 1 < nk) qa_wait_vmcnt<QA_RA + QA_RB>();
    else qa_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute(st);
    if (kb + 2 < nk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_tile(kb + 2, st);
    }
  }
  __syncthreads();                                  // the epilogue reuses the stage memory

  // ---- write-out of the projection: + bias, bf16.  Accumulator register r of a lane = row cr = (r & 3) + 8 (r >> 2)
  // + 4 fhalf of the 32-row block, column frow.  q / k: row-major [128][64], chunks swizzled by (row >> 1) & 7;
  // v: transposed [64 dims][128 rows], four consecutive rows of a lane as one 8-byte store.
  auto write_rm = [&](char* buf, const f32x16& a, int i, int colblk, float bias) {
    const int col = colblk * 32 + frow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      *reinterpret_cast<bf16_t*>(buf + row * 128 + (((col >> 3) ^ ((row >> 1) & 7)) << 4) + (col & 7) * 2) = (bf16_t)(a[r] + bias);
    }
  };
  auto write_vt = [&](const f32x16& a, int i, int colblk, float bias) {
    char* v0 = smem + QA_VT_OFF + (colblk * 32 + frow) * QA_VT_LD + (wm * 64 + i * 32 + 4 * fhalf) * 2;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                  // rows 8 g + 4 fhalf + (0..3) of the block
      const uint32_t lo = pack2(a[4 * g] + bias, a[4 * g + 1] + bias);
      const uint32_t hi = pack2(a[4 * g + 2] + bias, a[4 * g + 3] + bias);
      *reinterpret_cast<uint2*>(v0 + g * 16) = uint2{lo, hi};
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (wn == 0) {                                  // column blocks 0, 1 = q | 2 = k columns 0..31
      write_rm(smem + QA_Q_OFF, acc[i][0], i, 0, bcol[0]);
      write_rm(smem + QA_Q_OFF, acc[i][1], i, 1, bcol[1]);
      write_rm(smem + QA_K_OFF, acc[i][2], i, 0, bcol[2]);
    } else {                                        // 3 = k columns 32..63 | 4, 5 = v
      write_rm(smem + QA_K_OFF, acc[i][0], i, 1, bcol[0]);
      write_vt(acc[i][1], i, 0, bcol[1]);
      write_vt(acc[i][2], i, 1, bcol[2]);
    }
  }
  __syncthreads();

  // ---- attention: wave w takes pixels 2 w and 2 w + 1 (tile rows pr .. pr + 16 = the pixel's 16 frames), 16x16x32 MFMAs
  const int l15 = lane & 15, g4 = lane >> 4;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int pr = wave_u * 32 + pp * 16;
    const int row = pr + l15;
    const int sw = (row >> 1) & 7;
    const char* qrow = smem + QA_Q_OFF + row * 128;
    const char* krow = smem + QA_K_OFF + row * 128;
    f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                // S^T[key][query] = sum_d K[key][d] Q[query][d]
      const int c = ((ks * 4 + g4) ^ sw) << 4;
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(krow + c);
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qrow + c);
      st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qb, st, 0, 0, 0);
    }
    // lane: query l15, keys 4 g4 + r.  Softmax over the 16 keys: in-lane over r, across g4 by two swaps
    float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
    mx = qa_max_rows(mx);
    float e[4], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f((st[r] - mx) * p.scale_log2e); sum += e[r]; }
    sum = qa_sum_rows(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
    const uint32_t pk0 = pack2(e[0] * inv, e[1] * inv), pk1 = pack2(e[2] * inv, e[3] * inv);
    // P as the A operand of P.V (rows = queries, k = keys 8 g' .. +7; keys 16..31 of the 32-deep slice are zero):
    // lane (query, g' = 0) <- keys 0..3 (own) | 4..7 (lane + 16); (query, 1) <- 8..11 (lane + 16) | 12..15 (lane + 32)
    uint32_t a0, b0, a1, b1, lo, hi, c0, d0, c1, d1;
    qa_swap16(pk0, a0, b0);
    qa_swap16(pk1, a1, b1);
    qa_swap32(pk0, lo, hi);
    qa_swap16(hi, c0, d0);
    qa_swap32(pk1, lo, hi);
    qa_swap16(hi, c1, d1);
    const bool r0 = g4 == 0, r1 = g4 == 1;
    u32x4 pw;
    pw[0] = r0 ? a0 : (r1 ? c0 : 0u);
    pw[1] = r0 ? a1 : (r1 ? c1 : 0u);
    pw[2] = r0 ? b0 : (r1 ? d0 : 0u);
    pw[3] = r0 ? b1 : (r1 ? d1 : 0u);
    const bf16x8 pa = __builtin_bit_cast(bf16x8, pw);
    // O[query][d] = sum_key P[query][key] V[key][d]: B operand from v^T (lane: dim db*16 + l15, keys 8 g' .. +7 of the
    // pixel; g' >= 2 meets the zero half of P: it re-reads the valid half, never uninitialised bytes)
    const char* vt = smem + QA_VT_OFF + l15 * QA_VT_LD + (pr + 8 * (g4 & 1)) * 2;
    f32x4_t od[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(vt + db * 16 * QA_VT_LD);
      od[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // lane: dim db*16 + l15, queries 4 g4 + r -> bf16 over the pixel's q rows (dead: this wave alone read them, above)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = pr + 4 * g4 + r;
        const int d = db * 16 + l15;
        char* dst = smem + QA_Q_OFF + orow * 128 + (((d >> 3) ^ ((orow >> 1) & 7)) << 4) + (d & 7) * 2;
        *reinterpret_cast<bf16_t*>(dst) = (bf16_t)od[db][r];
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's 32 output rows are in LDS (written by this wave only)

  // ---- store: the wave's 32 rows x 64 columns, 8 lanes per row (128 contiguous bytes), 8 rows per pass
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int lr = wave_u * 32 + ps * 8 + (lane >> 3);
    const int ch = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + QA_Q_OFF + lr * 128 + ((ch ^ ((lr >> 1) & 7)) << 4));
    const int64_t m = row0 + (int64_t)(lr & 15) * p.hw + (lr >> 4);
    *reinterpret_cast<u32x4*>(p.out + m * p.ldo + h * 64 + ch * 8) = v;
  }
}

int qa_mode() {        // TC_QKV_ATTN = 0 never | 1 (default) wherever eligible and tb_fused.hip does not take the block; read per call
  const char* e = getenv("TC_QKV_ATTN");
  return e ? atoi(e) : 1;
}

}  // namespace

extern "C" int tc_temporal_qkv_attn_eligible(const TcTqaParams* p) {
  if (!p || qa_mode() == 0) return 0;
  if (p->t != QA_T || p->b <= 0 || p->hw <= 0 || (p->hw & 7)) return 0;
  if (p->heads <= 0 || p->c != p->heads * 64) return 0;
  if (p->ldx < p->c || p->ldo < p->c || (p->ldx & 7) || (p->ldo & 7)) return 0;
  // per-lane offsets are relative to the tile's first row and span 16 frames: 31-bit
  if (((int64_t)QA_T * p->hw + 8) * p->ldx * 2 >= 0x7fffff00LL) return 0;
  if ((int64_t)3 * p->c * p->c * 2 >= 0x7fffff00LL) return 0;
  const int64_t blocks = (int64_t)p->heads * 8 * (((int64_t)p->b * (p->hw / 8) + 7) / 8);
  if (blocks > 0x7fffffffLL) return 0;
  return 1;
}

extern "C" int tc_temporal_qkv_attn(const TcTqaParams* p, void* stream) {
  if (!p || !p->x || !p->wqkv || !p->out) return TC_EINVAL;
  if (!tc_temporal_qkv_attn_eligible(p)) return TC_ESHAPE;
  if (!tc_aligned16(p->x) || !tc_aligned16(p->wqkv) || !tc_aligned16(p->out)) return TC_EALIGN;
  QaArgs a;
  a.x = reinterpret_cast<const bf16_t*>(p->x); a.w = reinterpret_cast<const bf16_t*>(p->wqkv); a.bias = p->bqkv;
  a.out = reinterpret_cast<bf16_t*>(p->out);
  a.hw = p->hw; a.c = p->c; a.heads = p->heads; a.ldx = p->ldx; a.ldo = p->ldo;
  a.scale_log2e = p->scale * 1.44269504088896340736f;
  a.tiles_per_b = p->hw / 8;
  a.tiles = p->b * a.tiles_per_b;
  const unsigned grid = (unsigned)(p->heads * 8 * ((a.tiles + 7) / 8));
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(QA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  TC_LAUNCH_CHECK();
  return TC_OK;
}
