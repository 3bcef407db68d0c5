// Weight-stationary bf16 MFMA GEMM for the SHORT-K linear layers of UNet level 0 (K = 320), gfx950.
//
//   C[M, N] = epilogue( norm?(A)[M, 320] * W[N, 320]^T )        M = 81920 rows at B = 2, N = 320 | 960 | 2560
//
// Why another kernel (profiles/r02_clip_kernel_stats.txt): with K = 320 the tiled kernels of
// gemm.hip run FIVE K-steps per 128x128 tile, each paying one global->LDS latency behind a vmcnt(0) + barrier, then an
// fp32 tile transpose through LDS: 8-10 us of block lifetime for ~1 us of MFMA work, 0.12-0.21 of the MFMA roof, and for
// N = 320 (out-projections, proj_in/out: 59 launches per forward) also 2x off the HBM roof they should sit on.
// At K = 320 the WHOLE weight slab of a block fits the register file instead:
//   * one block per CU, 4 waves, persistent over a strided set of 64-row tiles; wave w owns NT*16 output columns
//     (80 for the plain layers: N-slab 320 = whole rows of the N = 320 projections; 64 for GEGLU: slab 256) and keeps
//     their W fragments for ALL of K in VGPRs (NT x 10 K-slices x 4 registers = 200 / 160 of the 512 a 1-wave-per-SIMD
//     kernel owns), loaded ONCE per block -- W never travels through LDS and is never re-read per tile;
//   * only A streams: whole-K 64 x 320 tiles (40 KiB) by LDS-DMA into a 3-deep ring, two tiles (80 KiB per CU, 20 MiB per
//     chip) in flight while one is consumed, counted vmcnt + ONE raw s_barrier per tile, no K loop at all;
//   * per tile a wave issues 4 x NT x 10 v_mfma_f32_16x16x32_bf16 from 40 ds_read_b128 (0.2 reads per MFMA);
//   * with A rows whole in LDS, LayerNorm (attention.py:225-227: the producer of every qkv / GEGLU input) becomes a
//     prologue on the tile: rows normalised in place ((x - mean) * rstd, fp32 statistics, two-pass variance), gamma / beta
//     folded into W / bias by the caller (exact in real arithmetic) -- the LayerNorm launch and its 2 x 52 MB disappear;
//   * GEGLU (attention.py:420-422) is computed IN the accumulator layout (value tile j and gate tile j+1 of a lane hold
//     the same (row, col) positions), so the fp32 tile transpose whose reads were 88 % bank conflicts is gone; what is
//     transposed through the per-wave slab is the finished product, half as many columns.
// Bytes: N = 320 reads A once and writes C once (+ residual once): HBM-bound by construction; N = 960 / 2560 re-read the
// 52 MB A from L2 / Infinity Cache per slab (blocks of one M-chunk share an XCD).
#include "gemm_common.h"

#include <stdlib.h>

namespace {

constexpr int WS_K = 320;
constexpr int WS_KS = WS_K / 32;                     // K-slices of one 16x16x32 MFMA
constexpr int WS_ROWS = 64;                          // rows per streamed A tile
constexpr int WS_MT = WS_ROWS / 16;
constexpr int WS_ROW_BYTES = WS_K * 2;               // 640 B = 40 chunks of 16 B
constexpr int WS_CHUNKS = WS_ROW_BYTES / 16;
constexpr int WS_TILE_BYTES = WS_ROWS * WS_ROW_BYTES;
constexpr int WS_STAGES = 3;
constexpr int WS_LOADS = WS_TILE_BYTES / 1024 / 4;   // LDS-DMA instructions per wave per tile (10)
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NT, bool GEGLU, bool RES, bool LN>
__global__ __launch_bounds__(256, 1) void gemm_ws_kernel(const TcGemmParams p, const int nchunks, const int safe_wait) {
  constexpr int WCOLS = NT * 16;                     // W rows (= accumulator columns) per wave
  constexpr int SLAB = 4 * WCOLS;                    // per block
  constexpr int OCOLS = GEGLU ? WCOLS / 2 : WCOLS;   // output columns per wave
  constexpr int VPR = OCOLS / 8;                     // 16-byte output vectors per row of the wave's slab
  constexpr int EPI_IT = (16 * VPR + 63) / 64;       // store instructions per 16-row pass
  constexpr int STORES = WS_MT * EPI_IT;             // per wave per tile (issued unconditionally: static count)
  constexpr int SLAB_FLOATS = 16 * OCOLS;
  static_assert(WS_LOADS + (RES ? 4 : 2) * STORES <= 63, "vmcnt is 6 bits");
  __shared__ __attribute__((aligned(1024))) char smem[WS_STAGES * WS_TILE_BYTES + 4 * SLAB_FLOATS * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fq = lane >> 4;

  // block -> (N-slab, M-chunk): all slabs of one chunk sit on one XCD (blockIdx & 7) and run side by side, so the
  // A tiles of the chunk are fetched into that L2 once
  const int slabs = p.n / SLAB;
  const int bid = blockIdx.x;
  const int chunk = (bid / (8 * slabs)) * 8 + (bid & 7);
  const int slab = (bid >> 3) % slabs;
  if (chunk >= nchunks) return;
  const int ntiles = (p.m + WS_ROWS - 1) / WS_ROWS;
  const int my_tiles = (ntiles - chunk + nchunks - 1) / nchunks;      // tiles chunk, chunk + nchunks, ...
  if (my_tiles <= 0) return;

  const tc_rsrc_t a_rsrc = make_rsrc(p.a, tc_a_extent(p));
  const tc_rsrc_t w_rsrc = make_rsrc(p.w, tc_w_extent(p));
  const int n_out = GEGLU ? p.n / 2 : p.n;
  const tc_rsrc_t c_rsrc = make_rsrc(p.c, ((int64_t)(p.m - 1) * p.ldc + n_out) * 2);

  // ---- W fragments of this wave, all of K, registers for the lifetime of the block.
  // v_mfma_f32_16x16x32_bf16 B operand: lane holds W row (lane & 15) of its 16-row tile, k = 8 (lane >> 4) .. +7 of the slice
  const int col_w0 = slab * SLAB + wave * WCOLS;
  bf16x8 wreg[NT][WS_KS];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const uint32_t voff = (uint32_t)((int64_t)(col_w0 + j * 16 + fr) * p.ldw * 2 + fq * 16);
#pragma unroll
    for (int ks = 0; ks < WS_KS; ++ks) wreg[j][ks] = __builtin_bit_cast(bf16x8, buf_load16(w_rsrc, voff, ks * 64));
  }
  float bias_r[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias_r[j] = p.bias ? p.bias[col_w0 + j * 16 + fr] : 0.f;

  // ---- A tile DMA geometry.  The tile is one contiguous 40 KiB LDS image [64 rows][40 chunks]; wave instruction
  // `inst` fills bytes [1024 inst, 1024 inst + 1024), lane l landing at linear chunk 64 inst + l = (row, position).
  // The position is the XOR-swizzled home of a logical chunk ((row >> 1) & 7 on the low three bits: conflict-free
  // ds_read_b128 fragment reads with 640-byte rows, as with the 128-byte rows of gemm.hip), applied on the SOURCE side.
  uint32_t a_voff[WS_LOADS];
  int a_row[WS_LOADS];
#pragma unroll
  for (int i = 0; i < WS_LOADS; ++i) {
    const int lin = (wave + 4 * i) * 64 + lane;
    const int row = lin / WS_CHUNKS, pos = lin - row * WS_CHUNKS;
    const int c = (pos & ~7) | ((pos & 7) ^ ((row >> 1) & 7));
    a_row[i] = row;
    a_voff[i] = (uint32_t)((int64_t)row * p.lda * 2 + c * 16);
  }
  const uint32_t tile_stride = (uint32_t)(WS_ROWS * p.lda * 2);
  // tiles past the end are still requested, with every row out of range (the hardware writes zeros): the number of
  // memory operations per iteration stays static, which the counted waits below rely on.  The row test is explicit
  // (TC_OOB in the per-lane offset): the scalar offset takes no part in it.
  auto load_tile = [&](int k, int stage) {
    const int tile = chunk + k * nchunks;
    const uint32_t base = (uint32_t)tile * tile_stride;            // < 2^31 for every tile that has a row < M (host check)
    char* dst = smem + stage * WS_TILE_BYTES + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < WS_LOADS; ++i)
      glds16(a_rsrc, dst + i * 4096, tile * WS_ROWS + a_row[i] < p.m ? base + a_voff[i] : TC_OOB, 0);
  };

  // fragment read offsets: row i*16 + fr, logical chunk 4 ks + fq -> position 8 (ks >> 1) + ((4 (ks & 1) + fq) ^ sw)
  const int sw = (fr >> 1) & 7;
  const int a_lane = fr * WS_ROW_BYTES;
  const int x_even = ((fq) ^ sw) << 4, x_odd = ((4 + fq) ^ sw) << 4;

  float* slab_f = reinterpret_cast<float*>(smem + WS_STAGES * WS_TILE_BYTES) + wave * SLAB_FLOATS;
  const int out_col_w0 = GEGLU ? col_w0 / 2 : col_w0;
  // epilogue vector geometry of this lane (per 16-row pass): EPI_IT vectors
  uint32_t e_off[EPI_IT];        // byte offset of the vector inside its row of C / residual
  int e_lds[EPI_IT];             // float offset inside the slab
  int e_row[EPI_IT];             // row inside the 16-row pass
  bool e_ok[EPI_IT];             // 16 * VPR vectors over 64 * EPI_IT slots: the surplus slots store nothing
#pragma unroll
  for (int q = 0; q < EPI_IT; ++q) {
    const int v = lane + 64 * q;
    e_ok[q] = v < 16 * VPR;
    const int lr = e_ok[q] ? v / VPR : 0, vc = e_ok[q] ? v - lr * VPR : 0;
    e_row[q] = lr;
    e_lds[q] = lr * OCOLS + vc * 8;
    e_off[q] = (uint32_t)((out_col_w0 + vc * 8) * 2);
  }
  const tc_rsrc_t r_rsrc = make_rsrc(RES ? p.residual : p.c, ((int64_t)(p.m - 1) * (RES ? p.ldr : p.ldc) + n_out) * 2);

  load_tile(0, 0);
  load_tile(1, 1);
  for (int t = 0; t < my_tiles; ++t) {
    // ---- tile t has landed (this wave's pieces), then every wave's (barrier).  Operations issued after its DMA:
    // t = 0: the DMA of tile 1; t >= 1: the previous iteration's DMA of tile t+1, residual loads and stores
    // (vmcnt retires in issue order; waiting for a smaller count than necessary is always safe)
    // Counting: the operations issued after the DMA of tile t are [residual loads (t-2), stores (t-2), DMA (t+1),
    // residual loads (t-1), stores (t-1)].  The default count lets the newest LOADS + (2 | 1) * STORES of them stay in
    // flight -- which also retires the stores of iteration t-2; safe_wait = 2 keeps those in flight as well (their
    // write acknowledgements can take longer than one ~1.5 us iteration: measured by scripts/ws_bench.py).
    if (safe_wait == 1) wait_vmcnt<0>();
    else if (t == 0) wait_vmcnt<WS_LOADS>();
    else if (safe_wait == 2 && t > 1) wait_vmcnt<WS_LOADS + (RES ? 4 : 2) * STORES>();
    else wait_vmcnt<WS_LOADS + (RES ? 2 : 1) * STORES>();
    __builtin_amdgcn_s_barrier();
    // the barrier also says: every wave is done reading stage (t + 2) % 3 (tile t - 1): refill it
    load_tile(t + 2, (t + 2) % WS_STAGES);
    const int tile = chunk + t * nchunks;
    char* sa = smem + (t % WS_STAGES) * WS_TILE_BYTES;

    if (LN) {
      // LayerNorm of the tile's rows in place (wave w: rows 16 w .. 16 w + 15; 4 lanes per row, 10 chunks each; the
      // statistics do not care about the chunk swizzle).  gamma / beta live in W / bias.
      const int row = wave * 16 + (lane >> 2);
      char* rp = sa + row * WS_ROW_BYTES + (lane & 3) * 16;
      float x[80];
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        unpack8(*reinterpret_cast<const u32x4*>(rp + q * 64), x + 8 * q);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[8 * q + e];
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float mean = s * (1.0f / WS_K);
      float v = 0.f;
#pragma unroll
      for (int e = 0; e < 80; ++e) { x[e] -= mean; v += x[e] * x[e]; }
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      const float rstd = rsqrtf(v * (1.0f / WS_K) + p.a_norm_eps);
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = x[8 * q + e] * rstd;
        *reinterpret_cast<u32x4*>(rp + q * 64) = pack8(y);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // every row of the tile is normalised before anyone multiplies it
    }

    // residual vectors of the whole tile: requested now, consumed after the MFMAs (latency under the matrix pipe)
    u32x4 rres[WS_MT][EPI_IT];
    if (RES) {
#pragma unroll
      for (int i = 0; i < WS_MT; ++i)
#pragma unroll
        for (int q = 0; q < EPI_IT; ++q) {
          const int m = tile * WS_ROWS + i * 16 + e_row[q];
          const uint32_t voff = (e_ok[q] && m < p.m) ? (uint32_t)((int64_t)m * p.ldr * 2) + e_off[q] : TC_OOB;
          rres[i][q] = buf_load16(r_rsrc, voff, 0);
        }
    }

    // ---- MFMAs: 10 K-slices x (4 A fragments from LDS) x (NT register-resident W fragments)
    f32x4_t acc[WS_MT][NT];
#pragma unroll
    for (int i = 0; i < WS_MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // A fragments double-buffered in registers: the four ds_read_b128 of slice ks + 1 are issued BEFORE the 4 x NT MFMAs
    // of slice ks (sched_barrier pins that order; left alone hipcc reads each fragment right before its use and waits
    // lgkmcnt(0) in front of every group of NT MFMAs)
    bf16x8 af[2][WS_MT];
    auto frags = [&](int ks, bf16x8 (&a)[WS_MT]) {
#pragma unroll
      for (int i = 0; i < WS_MT; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(sa + a_lane + i * (16 * WS_ROW_BYTES) + (ks >> 1) * 128 +
                                                ((ks & 1) ? x_odd : x_even));
    };
    frags(0, af[0]);
#pragma unroll
    for (int ks = 0; ks < WS_KS; ++ks) {
      if (ks + 1 < WS_KS) frags(ks + 1, af[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WS_MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1][i], wreg[j][ks], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: four 16-row passes through the wave's private fp32 slab (C/D layout of the 16x16 MFMA:
    // col = lane & 15, row = 4 (lane >> 4) + reg); LDS operations of one wave complete in order
#pragma unroll
    for (int i = 0; i < WS_MT; ++i) {
      if (GEGLU) {
#pragma unroll
        for (int jj = 0; jj < NT / 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            slab_f[(fq * 4 + r) * OCOLS + jj * 16 + fr] =
                (acc[i][2 * jj][r] + bias_r[2 * jj]) * gelu_erf_f(acc[i][2 * jj + 1][r] + bias_r[2 * jj + 1]);
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab_f[(fq * 4 + r) * OCOLS + j * 16 + fr] = acc[i][j][r] + bias_r[j];
      }
#pragma unroll
      for (int q = 0; q < EPI_IT; ++q) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab_f + e_lds[q]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab_f + e_lds[q] + 4);
        float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (RES) {
          float rf[8];
          unpack8(rres[i][q], rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += rf[e];
        }
        const int m = tile * WS_ROWS + i * 16 + e_row[q];
        const uint32_t voff = (e_ok[q] && m < p.m) ? (uint32_t)((int64_t)m * p.ldc * 2) + e_off[q] : TC_OOB;   // dropped
        __builtin_amdgcn_raw_buffer_store_b128(pack8(x), c_rsrc, voff, 0, 0);
      }
    }
  }
  wait_vmcnt<0>();        // the two run-ahead DMA requests target this block's LDS: drain them before it is released
}

int ws_mode() {            // TC_GEMM_WS = 0 never | 1 heuristic (default) | 2 whenever the shape allows | 3 = 2 + vmcnt(0) waits | 4 = 2 + deeper store window | 5 = 1 + deeper store window
  const char* e = getenv("TC_GEMM_WS");      // read per call: the parity tests flip it inside one process
  return e ? atoi(e) : 1;
}

}  // namespace

// Would the weight-stationary kernel take this (validated) problem?  One definition for the launcher and for the host
// layer that decides whether a LayerNorm may be folded into its consumer (tc_gemm_ws_eligible).
static bool ws_shape_ok(const TcGemmParams& p, int batch, int mode) {
  // TC_GEMM_TILE forces a tile family: tc_gemm_bf16 then never comes here, so the host must not be told that a
  // LayerNorm may be folded into this launch (the two rules disagreed under forced-tile sweeps)
  if (const char* e = getenv("TC_GEMM_TILE")) { if (e[0]) return false; }
  if (mode == 0 || batch != 1 || p.gather != TC_GATHER_LINEAR || p.k != WS_K || p.lda < WS_K) return false;
  if (p.row_bias || p.alpha != 1.f || p.out_scale != 1.f || p.out_f32) return false;
  const bool geglu = p.act == TC_ACT_GEGLU;
  if (!geglu && p.act != TC_ACT_NONE) return false;
  if (p.n % (geglu ? 256 : 320) != 0) return false;
  if (p.residual && geglu) return false;
  // Heuristic (profiles/r03_ws_bench_hipblaslt_yardstick.txt, B = 2 shapes): the N = 320 projections are HBM-bound and gain
  // 1.15x (plain / + residual) to 1.53x (LayerNorm prologue instead of a LayerNorm launch); with more than one N-slab the
  // single wave per SIMD serialises LayerNorm, MFMAs and epilogue and the kernel LOSES to the tiled one (qkv 0.89x,
  // 0.67x with the prologue; GEGLU 0.61x / 0.50x), and at M = 40960 it only ties -- so: one slab, plain, M >= 64K rows.
  if ((mode == 1 || mode == 5) && (p.m < 65536 || geglu || p.n != 320)) return false;
  if ((int64_t)p.m * p.lda * 2 >= 0x7fffff00LL) return false;       // this kernel addresses A from the tensor base
  if ((int64_t)p.m * p.ldc * 2 >= 0x7fffff00LL || (p.residual && (int64_t)p.m * p.ldr * 2 >= 0x7fffff00LL)) return false;
  return true;
}

extern "C" int tc_gemm_ws_eligible(const TcGemmParams* p) {
  if (!p) return 0;
  return ws_shape_ok(*p, p->batch > 0 ? p->batch : 1, ws_mode()) ? 1 : 0;
}

// 1 = launched
int tc_gemm_ws_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry) {
  const int mode = ws_mode();
  if (!ws_shape_ok(p, batch, mode)) return 0;
  if (dry) return 1;
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int slabs = p.n / (geglu ? 256 : 320);
  const int ntiles = (p.m + WS_ROWS - 1) / WS_ROWS;
  int nchunks = 256 / slabs;                                    // one block per CU
  if (nchunks < 1) nchunks = 1;
  if (nchunks > ntiles) nchunks = ntiles;
  const int grid = slabs * 8 * ((nchunks + 7) / 8);
  // wait mode: the heuristic (1) runs with the deeper store window (+1-6 %, profiles/r03_ws_bench_hipblaslt_yardstick.txt);
  // 2 = counted waits that also retire the stores of two tiles ago, 3 = vmcnt(0) everywhere (parity tests)
  const int safe = mode == 3 ? 1 : (mode == 2 ? 0 : 2);
  const bool res = p.residual != nullptr, ln = p.a_norm != 0;
  dim3 g((unsigned)grid), b(256);
#define TC_WS_LAUNCH(NT, G, R, L) hipLaunchKernelGGL((gemm_ws_kernel<NT, G, R, L>), g, b, 0, s, p, nchunks, safe)
  if (geglu) { if (ln) TC_WS_LAUNCH(4, true, false, true); else TC_WS_LAUNCH(4, true, false, false); }
  else if (res) { if (ln) TC_WS_LAUNCH(5, false, true, true); else TC_WS_LAUNCH(5, false, true, false); }
  else { if (ln) TC_WS_LAUNCH(5, false, false, true); else TC_WS_LAUNCH(5, false, false, false); }
#undef TC_WS_LAUNCH
  return 1;
}
