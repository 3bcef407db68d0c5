// GroupNorm(32) (+SiLU), LayerNorm and row softmax over channels-last bf16 rows, gfx950.
// All three are HBM-bound: 16-byte vector loads/stores, fp32 statistics, wave64 shuffles.
//
// GroupNorm is three launches:
//   gn_stats    : grid (chunks, samples).  A block walks its chunk of rows with every thread
//                 pinned to one 8-channel vector (so per-channel sums stay in registers, four
//                 row loads in flight per thread), folds channels -> groups through LDS and
//                 writes one (sum, sumsq) pair per (sample, chunk, group).  No atomics:
//                 bit-reproducible.
//   gn_finalize : grid (samples).  256 threads reduce the per-chunk partials of the 32 groups
//                 (8 lanes per group + shuffles) to (mean, rstd).
//   gn_apply    : same decomposition as gn_stats; folds mean/rstd with gamma/beta into
//                 per-channel (scale, shift) registers and streams y = silu(x * scale + shift).
// (Folding gn_finalize into gn_stats -- last block of a sample reduces, arrival counter + agent-scope
// __threadfence() in every block -- was built and measured: the 2048 L2 write-back/invalidate fences
// per launch cost far more than the 5 us launch they save, +27 % on the whole clip.  Folding it into every
// gn_apply block's prologue instead (partials re-reduced per block, no fences) measured neutral under hipGraph
// replay.  Three launches stay.)
// The chunk height is chosen on the host so that ~512 blocks are in flight whatever the
// tensor shape (clip-wide statistics have only B samples, per-frame ones B*T).
// Algorithmic traffic: read x twice, write y once (the second read mostly hits the 256 MiB
// Infinity Cache for UNet-sized tensors).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SLOTS = 2;       // 8-channel vectors per thread per row pass -> C <= 4096
constexpr int GN_TARGET_BLOCKS = 512;     // 2 per CU: measured best of 128..8192 (profiles/r01_v6_norm_bench.txt)
constexpr int GN_MIN_ROWS = 8;

// ---- weight prefetch planes (ABI 12: tc_groupnorm_pf / tc_layernorm_pf) ------------------------------------------------
// Inside the UNet forward every weight matrix is read once per forward -- 2.9 GB of weights cycle through a 256 MB Infinity
// Cache, so a GEMM's W always comes from HBM, while a per-shape microbenchmark has it warm.  Measured
// (profiles/r05_cold_operand_probe.txt): that costs the 1280-channel layers 7-29 % of their time (level 3: 16-29 %, level-2
// 3x3 convolutions and projections ~10 %), nothing at levels 0 / 1 where W is small beside A -- and a read of W by ANOTHER
// kernel one launch earlier takes 93-100 % of it away (profiles/r05_prefetch_premise_probe.txt).  The norm in front of a
// GEMM is that other kernel: it is latency-bound at these sizes (10-15 us for 3-26 MB), so extra blocks that stream the
// consumer's weights through a load and drop them ride in its shadow.  The extra blocks are whole grid PLANES (blockIdx.z >= 1
// for GroupNorm, blockIdx.y >= 1 for LayerNorm): the norm's own blocks do not change.
constexpr int PF_MAX = 4;
struct PfArgs {
  const u32x4* p[PF_MAX];
  uint32_t units[PF_MAX];        // 16-byte units
  int n;
};

// block b of nb prefetch blocks: its share of every tensor, 16 loads of 16 bytes per thread in flight, values dropped
__device__ __forceinline__ void pf_run(const PfArgs& pf, uint32_t b, uint32_t nb, uint32_t tid, uint32_t nthr) {
#pragma unroll 1
  for (int i = 0; i < pf.n; ++i) {
    const u32x4* src = pf.p[i];
    const uint32_t units = pf.units[i];
    uint32_t per = (units + nb - 1) / nb;
    per = (per + 7u) & ~7u;                                   // whole 128-byte lines per block
    const uint32_t lo = b * per;
    if (lo >= units) continue;
    const uint32_t hi = min(units, lo + per);
#pragma unroll 1
    for (uint32_t u = lo + tid; u < hi; u += nthr * 16u) {
      u32x4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t uu = u + (uint32_t)k * nthr;
        v[k] = uu < hi ? src[uu] : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k][0]), "v"(v[k][1]), "v"(v[k][2]), "v"(v[k][3]));
    }
  }
}

struct GnGeo {
  int vpr;         // 8-channel vectors per row (C/8)
  int slots;       // vector slots per thread (1 or 2)
  int rows_pp;     // rows processed per pass by one block
};

__host__ __device__ __forceinline__ GnGeo gn_geo(int c) {
  GnGeo g;
  g.vpr = c >> 3;
  g.slots = (g.vpr + GN_THREADS - 1) / GN_THREADS;
  const int vps = (g.vpr + g.slots - 1) / g.slots;   // vectors per slot-pass
  g.rows_pp = g.slots > 1 ? 1 : GN_THREADS / vps;
  if (g.rows_pp < 1) g.rows_pp = 1;
  return g;
}

// thread -> (row lane, vector index for slot s); -1 when idle
__device__ __forceinline__ void gn_thread_map(const GnGeo& g, int tid, int& rlane, int (&vec)[GN_MAX_SLOTS]) {
  if (g.slots > 1) {
    rlane = 0;
#pragma unroll
    for (int s = 0; s < GN_MAX_SLOTS; ++s) {
      const int v = tid + s * GN_THREADS;
      vec[s] = (s < g.slots && v < g.vpr) ? v : -1;
    }
  } else {
    rlane = tid / g.vpr;
    vec[0] = rlane < g.rows_pp ? tid - rlane * g.vpr : -1;
    if (rlane >= g.rows_pp) rlane = 0;
    vec[1] = -1;
  }
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                              int rows, int c, int nchunks, int chunk_rows) {
  __shared__ float red[2][4096 + 32];
  const GnGeo g = gn_geo(c);
  const int tid = threadIdx.x;
  int rlane, vec[GN_MAX_SLOTS];
  gn_thread_map(g, tid, rlane, vec);
  const int sample = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * chunk_rows;
  const int r1 = min(rows, r0 + chunk_rows);
  const bf16_t* xs = x + (int64_t)sample * rows * c;

  float sum[GN_MAX_SLOTS][8], sq[GN_MAX_SLOTS][8];
#pragma unroll
  for (int s = 0; s < GN_MAX_SLOTS; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[s][e] = 0.f; sq[s][e] = 0.f; }

  const int step = g.rows_pp;
  for (int r = r0 + rlane; r < r1; r += 4 * step) {
#pragma unroll
    for (int s = 0; s < GN_MAX_SLOTS; ++s) {
      if (vec[s] >= 0) {
        u32x4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * step;
          v4[u] = rr < r1 ? *reinterpret_cast<const u32x4*>(xs + (int64_t)rr * c + vec[s] * 8)
                          : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[8];
          unpack8(v4[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { sum[s][e] += f[e]; sq[s][e] += f[e] * f[e]; }
        }
      }
    }
  }
  // fold row lanes: per-channel totals in LDS (row lane 0 initialises, the others add in turn)
  for (int pass = 0; pass < g.rows_pp; ++pass) {
    if (rlane == pass) {
#pragma unroll
      for (int s = 0; s < GN_MAX_SLOTS; ++s)
        if (vec[s] >= 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ch = vec[s] * 8 + e;
            if (pass == 0) { red[0][ch] = sum[s][e]; red[1][ch] = sq[s][e]; }
            else { red[0][ch] += sum[s][e]; red[1][ch] += sq[s][e]; }
          }
        }
    }
    __syncthreads();
  }
  if (tid < 32) {
    const int cpg = c / 32;
    float a = 0.f, b = 0.f;
    for (int i = 0; i < cpg; ++i) { a += red[0][tid * cpg + i]; b += red[1][tid * cpg + i]; }
    float* o = part + (((int64_t)sample * nchunks + chunk) * 32 + tid) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// part: [samples][nchunks][32][2] -> stats: [samples][32][2] = (mean, rstd).
// One wave per (sample, group): 64 lanes stride over the chunks, then a wave reduction.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                          int samples, int rows, int c, int nchunks, float eps) {
  const int lane = threadIdx.x & 63;
  const int sg = blockIdx.x * 4 + (threadIdx.x >> 6);        // sample*32 + group
  if (sg >= samples * 32) return;
  const int sample = sg >> 5, grp = sg & 31;
  const float* pp = part + ((int64_t)sample * nchunks * 32 + grp) * 2;
  double a = 0.0, b = 0.0;
  for (int k = lane; k < nchunks; k += 64) {
    const float2 v = *reinterpret_cast<const float2*>(pp + (int64_t)k * 64);
    a += v.x;
    b += v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  if (lane == 0) {
    const double cnt = (double)rows * (c / 32);
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(int64_t)sg * 2] = (float)mean;
    stats[(int64_t)sg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ABI 9: (mean, rstd) from a PRODUCER's partial sums (TcGemmParams.gn_part: per block of `prows` rows and per channel
// the sum and the sum of squares of the values it stored): part [blocks][2][c] -> stats [samples][32][2].  One 256-thread
// block per (sample, group): thread t takes row blocks t, t + 256, ... and walks the group's cpg contiguous channels
// (clip-wide norms have only 2 x 32 (sample, group) pairs but 256 row blocks each: a wave per pair -- the first version --
// spent 40 dependent round trips there and made the whole operator SLOWER than the two-pass one); fp64, fixed order.
__global__ __launch_bounds__(256) void gn_finalize_part_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                               int samples, int rows, int c, int prows, float eps) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sg = blockIdx.x;                                  // sample*32 + group
  const int sample = sg >> 5, grp = sg & 31;
  const int cpg = c / 32, nb = rows / prows;
  const float* pp = part + ((int64_t)sample * nb * 2) * c + grp * cpg;
  double a = 0.0, b = 0.0;
  for (int blk = tid; blk < nb; blk += 256) {
    const float* r0 = pp + (int64_t)blk * 2 * c;
    for (int ch = 0; ch < cpg; ++ch) { a += r0[ch]; b += r0[c + ch]; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (tid == 0) {
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double cnt = (double)rows * cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(int64_t)sg * 2] = (float)mean;
    stats[(int64_t)sg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// (Round 6: the finalize launch folded into THIS kernel's prologue -- every apply block reduces its sample's partial sums itself,
// thread (pair, quarter) over chunks quarter, quarter + 4, ... in fp64, fixed order: bit-identical to the three-launch form on
// every shape tried -- removes 94 of a forward's 923 launches and LOSES 2.4 % per guided forward (35.27 -> 36.15 ms; the decoder's
// norms 10.0 -> 10.7 ms): up to 64 KB of partials read and reduced by each of ~512 blocks costs twice the 4.6 us launch it
// saves.  profiles/r06_gn_finalize_in_apply_forward_ab.txt.  Removed; round 1 had measured the same idea "neutral".)
// SILU is a template parameter: as a run-time flag every element paid for the activation AND a select
template <bool SILU>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, int rows, int c,
                                                              int chunk_rows, const PfArgs pf) {
  if (blockIdx.z) {                                           // a weight-prefetch plane (ABI 12)
    pf_run(pf, ((blockIdx.z - 1) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, (gridDim.z - 1) * gridDim.y * gridDim.x,
           threadIdx.x, GN_THREADS);
    return;
  }
  const GnGeo g = gn_geo(c);
  const int tid = threadIdx.x;
  int rlane, vec[GN_MAX_SLOTS];
  gn_thread_map(g, tid, rlane, vec);
  const int sample = blockIdx.y, chunk = blockIdx.x;
  const float* st = stats + (int64_t)sample * 64;
  const int cpg = c / 32;
  float sc[GN_MAX_SLOTS][8], sh[GN_MAX_SLOTS][8];
#pragma unroll
  for (int s = 0; s < GN_MAX_SLOTS; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[s][e] = 0.f; sh[s][e] = 0.f;
      if (vec[s] >= 0) {
        const int ch = vec[s] * 8 + e;
        const int grp = ch / cpg;
        const float a = st[grp * 2 + 1] * gamma[ch];
        sc[s][e] = a;
        sh[s][e] = beta[ch] - st[grp * 2] * a;
      }
    }
  const int r0 = chunk * chunk_rows;
  const int r1 = min(rows, r0 + chunk_rows);
  const bf16_t* xs = x + (int64_t)sample * rows * c;
  bf16_t* ys = y + (int64_t)sample * rows * c;
  const int step = g.rows_pp;
  for (int r = r0 + rlane; r < r1; r += 4 * step) {
#pragma unroll
    for (int s = 0; s < GN_MAX_SLOTS; ++s) {
      if (vec[s] >= 0) {
        u32x4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * step;
          if (rr < r1) v4[u] = *reinterpret_cast<const u32x4*>(xs + (int64_t)rr * c + vec[s] * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * step;
          if (rr < r1) {
            float f[8];
            unpack8(v4[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = f[e] * sc[s][e] + sh[s][e];
              f[e] = SILU ? silu_f(v) : v;
            }
            *reinterpret_cast<u32x4*>(ys + (int64_t)rr * c + vec[s] * 8) = pack8(f);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Single-pass GroupNorm for slabs that fit a block's registers.  A block owns one (sample, unit) where a unit is
// U = lcm(8, C/32) channels (whole 8-channel vectors AND whole groups: 40 channels = 4 groups at C = 320, 2 at 640,
// 1 at 1280; 120 at C = 960 / 1920; 80 at 2560).  Every thread is pinned to one 8-channel vector of the unit and walks
// the rows, keeping its NV vectors (packed bf16) in registers: x is read from memory ONCE (the three-launch path reads
// it twice), the variance is the true two-pass sum((x - mean)^2) (no E[x^2] - mean^2 cancellation), and the whole norm
// is one launch instead of three -- for the level-2/3 tensors of the UNet (3-13 MB) the launches, not the bytes, were
// the cost (~15 us for 1-4 us of traffic).  Reductions: per-lane group contributions -> wave shuffles -> one LDS slot
// per (wave, group) -> summed in wave order by every thread: no atomics, bit-reproducible.
// keeps the compiler from carrying the UNPACKED fp32 copies of the slab from one pass to the next (8 registers per
// vector instead of 4: the 26-vector instance would spill): after this the packed words are "new" values
__device__ __forceinline__ void gn_opaque(u32x4& v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }

template <int T, int NV, bool SILU>
__global__ __launch_bounds__(T) void gn_onepass_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int rows, int c, int vu, float eps, const PfArgs pf) {
  if (blockIdx.z) {                                           // a weight-prefetch plane (ABI 12)
    pf_run(pf, ((blockIdx.z - 1) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, (gridDim.z - 1) * gridDim.y * gridDim.x,
           threadIdx.x, T);
    return;
  }
  constexpr int NW = T / 64;
  __shared__ float red[2][NW][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cpg = c >> 5;
  const int rpp = T / vu;                               // rows per pass; threads >= rpp * vu idle
  const int col = tid % vu, r0 = tid / vu;
  const bool live = tid < rpp * vu;
  const int unit = blockIdx.x, sample = blockIdx.y;
  const int ch0 = (unit * vu + col) * 8;                // first channel of this thread's vector
  const int ug0 = (unit * vu * 8) / cpg;                // first group of the unit
  const bf16_t* xs = x + (int64_t)sample * rows * c + ch0;
  bf16_t* ys = y + (int64_t)sample * rows * c + ch0;
  int gl[8];                                            // group (local to the unit, 0..3) of each of the 8 channels
#pragma unroll
  for (int e = 0; e < 8; ++e) gl[e] = (ch0 + e) / cpg - ug0;

  u32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int r = r0 + i * rpp;
    v[i] = (live && r < rows) ? *reinterpret_cast<const u32x4*>(xs + (int64_t)r * c) : u32x4{0u, 0u, 0u, 0u};
  }
  // block-wide sum of a per-channel quantity, per group: lane -> 4 group slots -> wave -> LDS -> everyone
  auto group_sums = [&](const float (&pc)[8], int slot, float (&out)[4]) {
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) g4[g] += (gl[e] == g) ? pc[e] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) g4[g] = wave_sum(g4[g]);
    if (lane == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) red[slot][wave][g] = g4[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[slot][w][g];
      out[g] = t;
    }
  };
  float pc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float f[8];
    unpack8(v[i], f);                                   // rows beyond the slab hold zeros: they add nothing
#pragma unroll
    for (int e = 0; e < 8; ++e) pc[e] += f[e];
  }
  float gs[4];
  group_sums(pc, 0, gs);
  const float inv_cnt = 1.0f / ((float)rows * (float)cpg);
  float mean_e[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) m = (gl[e] == g) ? gs[g] * inv_cnt : m;
    mean_e[e] = m;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) pc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int r = r0 + i * rpp;
    gn_opaque(v[i]);
    if (live && r < rows) {
      float f[8];
      unpack8(v[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean_e[e]; pc[e] += d * d; }
    }
  }
  float gq[4];
  group_sums(pc, 1, gq);
  float sc[8], sh[8];
  if (live) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch0), g1 = *reinterpret_cast<const f32x4*>(gamma + ch0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch0), b1 = *reinterpret_cast<const f32x4*>(beta + ch0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float var = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) var = (gl[e] == g) ? gq[g] * inv_cnt : var;
      const float a = rsqrtf(var + eps) * (e < 4 ? g0[e] : g1[e - 4]);
      sc[e] = a;
      sh[e] = (e < 4 ? b0[e] : b1[e - 4]) - mean_e[e] * a;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int r = r0 + i * rpp;
      gn_opaque(v[i]);
      if (r < rows) {
        float f[8];
        unpack8(v[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = f[e] * sc[e] + sh[e];
          f[e] = SILU ? silu_f(t) : t;
        }
        *reinterpret_cast<u32x4*>(ys + (int64_t)r * c) = pack8(f);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// LayerNorm: a wave normalises R rows at a time, each row held in registers (NV vectors of 8 per lane), true
// two-pass variance.  R > 1 for the narrow rows (C <= 512: R = 4, C <= 1024: R = 2) puts several row loads in
// flight per wave -- at C = 320 only 40 of the 64 lanes carry data, so one row per wave left the kernel
// latency-bound (4.1 TB/s).
// MX = true (tc_layernorm_mxfp8): the normalised row leaves as MXFP8 -- e4m3 bytes q[row, ldq] plus one E8M0 scale per
// 32 channels s[row, lds] -- instead of bf16: exactly tc_quant_mxfp8 applied to the bf16 result (the value is rounded
// to bf16 first), without the bf16 round trip through HBM.  A 32-channel block is 4 adjacent lanes.
template <int NV, int R, bool MX = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int rows, int c, float eps, uint8_t* __restrict__ q = nullptr,
                                                       int ldq = 0, uint8_t* __restrict__ sc = nullptr, int lds = 0,
                                                       const PfArgs pf = PfArgs{}) {
  if (blockIdx.y) {                                           // a weight-prefetch plane (ABI 12)
    pf_run(pf, (blockIdx.y - 1) * gridDim.x + blockIdx.x, (gridDim.y - 1) * gridDim.x, threadIdx.x, 256);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int vpr = c >> 3;
  float f[R][NV][8];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = min(row0 + r, rows - 1);                 // clamped rows are recomputed, never stored
    const bf16_t* xr = x + (int64_t)row * c;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = lane + i * 64;
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (v < vpr) raw = *reinterpret_cast<const u32x4*>(xr + v * 8);
      unpack8(raw, f[r][i]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[r] += f[r][i][e];        // lanes beyond the row hold zeros
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)c;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = lane + i * 64;
      if (v < vpr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[r][i][e] - mean[r]; q += d * d; }
      }
    }
    rstd[r] = rsqrtf(wave_sum(q) / (float)c + eps);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = lane + i * 64;
    if (v < vpr) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (row0 + r < rows) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f[r][i][e] - mean[r]) * rstd[r] * g0[e] + b0[e];
            o[4 + e] = (f[r][i][4 + e] - mean[r]) * rstd[r] * g1[e] + b1[e];
          }
          const u32x4 packed = pack8(o);
          if (!MX) {
            *reinterpret_cast<u32x4*>(y + (int64_t)(row0 + r) * c + v * 8) = packed;
          } else {
            uint32_t amax = 0;                               // |value| as bf16 bits, as in quant_mx_kernel
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = max(amax, max(packed[e] & 0x7fffu, (packed[e] >> 16) & 0x7fffu));
            amax = max(amax, (uint32_t)__shfl_xor((int)amax, 1, 64));
            amax = max(amax, (uint32_t)__shfl_xor((int)amax, 2, 64));
            const int e8 = (int)(amax >> 7);
            const int byte = e8 - 8 < 0 ? 0 : (e8 - 8 > 254 ? 254 : e8 - 8);
            const float inv = __uint_as_float((uint32_t)(254 - byte) << 23);
            float t[8];
            unpack8(packed, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf(t[e] * inv, -448.f), 448.f);
            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w0, true);
            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], w1, true);
            *reinterpret_cast<u32x2*>(q + (int64_t)(row0 + r) * ldq + v * 8) = u32x2{(uint32_t)w0, (uint32_t)w1};
            if ((v & 3) == 0) sc[(int64_t)(row0 + r) * lds + (v >> 2)] = (uint8_t)byte;
          }
        }
      }
    } else if (MX && v - vpr < lds - (c >> 5)) {
      // padding columns of the scale matrix (K not a multiple of 128) are written as zeros by the first idle lanes
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (row0 + r < rows) sc[(int64_t)(row0 + r) * lds + (c >> 5) + (v - vpr)] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row softmax fp32 -> bf16, one block per row (rows are L = 2560 wide in the decoder).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                          int n_all, int n_out, int lds, int ldo, int causal_period) {
  __shared__ float redm[4], reds[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sr = s + (int64_t)blockIdx.x * lds;
  bf16_t* pr = p + (int64_t)blockIdx.x * ldo;
  int n = n_all;
  if (causal_period > 0) n = min(n_all, (int)(blockIdx.x % causal_period) + 1);
  float mx = -1e30f;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, sr[i]);
  mx = wave_max(mx);
  if (lane == 0) redm[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float sum = 0.f;
  for (int i = tid; i < n; i += 256) sum += __expf(sr[i] - mx);
  sum = wave_sum(sum);
  if (lane == 0) reds[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (reds[0] + reds[1] + reds[2] + reds[3]);
  for (int i = tid; i < n_out; i += 256) pr[i] = i < n ? (bf16_t)(__expf(sr[i] - mx) * inv) : (bf16_t)0.f;
}

}  // namespace

// ABI 12: the caller's prefetch list -> kernel arguments.  Returns a TC_E* code; pf == nullptr or n == 0: nothing to prefetch.
static int pf_args(const TcPrefetch* pf, PfArgs* out) {
  PfArgs a{};
  a.n = 0;
  if (pf) {
    if (pf->n < 0 || pf->n > TC_PREFETCH_MAX) return TC_EINVAL;
    for (int i = 0; i < pf->n; ++i) {
      if (!pf->ptr[i] || pf->bytes[i] < 0 || pf->bytes[i] > ((int64_t)1 << 35)) return TC_EINVAL;
      if (!tc_aligned16(pf->ptr[i])) return TC_EALIGN;
      const uint32_t units = (uint32_t)(pf->bytes[i] >> 4);      // whole 16-byte units only: never a byte beyond the tensor
      if (!units) continue;
      a.p[a.n] = reinterpret_cast<const u32x4*>(pf->ptr[i]);
      a.units[a.n] = units;
      ++a.n;
    }
  }
  *out = a;
  return TC_OK;
}
// planes of prefetch blocks beside a grid plane of `plane` blocks: at least ~512 blocks (one 16-deep pass of 256 threads
// covers 64 KiB: 512 blocks = 32 MiB per pass) in at most 8 planes -- unless the norm's own plane is so small (a one-pass
// GroupNorm of a few (sample, unit) slabs, a LayerNorm over a few rows) that 8 planes would leave a handful of blocks
// streaming megabytes each, serially, as the launch's critical path: then as many planes as give every 64 KiB
// of the list its own block, up to 512 blocks / 64 planes.  The block count follows the BYTES, not the norm's size.
static inline unsigned pf_planes(const PfArgs& a, int64_t plane) {
  if (!a.n || plane <= 0) return 0;
  int64_t z = (512 + plane - 1) / plane;
  if (z > 8) {
    int64_t units = 0;
    for (int i = 0; i < a.n; ++i) units += a.units[i];
    int64_t need = (units * 16 + 65535) / 65536;             // blocks of one 16-deep pass
    if (need > 512) need = 512;
    const int64_t zn = (need + plane - 1) / plane;
    z = zn > 8 ? (zn > 64 ? 64 : zn) : 8;
  }
  return (unsigned)(z < 1 ? 1 : z);
}

// chunk height: ~GN_TARGET_BLOCKS blocks in flight, at least GN_MIN_ROWS rows each
static inline void gn_chunking(int samples, int rows, int* nchunks, int* chunk_rows) {
  static const int target = [] { const char* e = getenv("TC_GN_BLOCKS"); const int v = e ? atoi(e) : 0;
                                 return v > 0 ? v : GN_TARGET_BLOCKS; }();       // tuning override
  int per_sample = target / (samples > 0 ? samples : 1);
  if (per_sample < 1) per_sample = 1;
  int max_chunks = (rows + GN_MIN_ROWS - 1) / GN_MIN_ROWS;
  int n = per_sample < max_chunks ? per_sample : max_chunks;
  if (n < 1) n = 1;
  const int cr = (rows + n - 1) / n;
  *chunk_rows = cr;
  *nchunks = (rows + cr - 1) / cr;
}

extern "C" int64_t tc_groupnorm_workspace(int32_t samples, int32_t rows, int32_t c) {
  (void)c;
  if (samples <= 0 || rows <= 0) return 0;
  int nch, cr;
  gn_chunking(samples, rows, &nch, &cr);
  return ((int64_t)samples * nch * 64 + (int64_t)samples * 64) * sizeof(float);
}

static int groupnorm_impl(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                          int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                          void* workspace, int64_t workspace_bytes, const TcPrefetch* prefetch, void* stream) {
  if (!x || !y || !gamma || !beta || !workspace || samples <= 0 || rows <= 0 || c <= 0) return TC_EINVAL;
  PfArgs pf;
  if (const int rc = pf_args(prefetch, &pf)) return rc;
  if ((c % 32) != 0 || c > GN_MAX_SLOTS * GN_THREADS * 8 || c > 4096) return TC_ESHAPE;
  if ((c % 8) != 0) return TC_ESHAPE;
  if (!tc_aligned16(x) || !tc_aligned16(y)) return TC_EALIGN;
  if (workspace_bytes < tc_groupnorm_workspace(samples, rows, c)) return TC_EWORKSPACE;
  if (samples > 65535) return TC_ESHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    // single-pass kernel when a (sample, unit) slab fits the registers of one block (TC_GN_ONEPASS=0: never)
    static const bool onepass = [] { const char* e = getenv("TC_GN_ONEPASS"); return !(e && e[0] == '0'); }();
    const int cpg = c / 32;
    int u = cpg;
    while (u % 8) u += cpg;                                       // lcm(8, cpg) = U channels per unit
    const int vu = u / 8, gu = u / cpg;
    if (onepass && (c % u) == 0 && gu <= 4 && vu <= 64 && tc_aligned16(gamma) && tc_aligned16(beta)) {
      const int64_t nvec = (int64_t)rows * vu;
      const dim3 grid(c / u, samples, 1 + pf_planes(pf, (int64_t)(c / u) * samples));
      const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
      bf16_t* yb = reinterpret_cast<bf16_t*>(y);
      auto fits = [&](int t, int nv) { return (int64_t)((rows + t / vu - 1) / (t / vu)) <= nv; };
      // measured (profiles/r02_gn_onepass_ab.txt): wins 13-32 % where the grid fills the chip (per-frame norms of levels
      // 1-3) or the tensor is tiny (level-3 clip-wide, 3 MB); LOSES with 64 blocks on a 13 MB tensor (level-2
      // clip-wide: 20 -> 31 us) and is neutral at level 0 (a 512-thread / 26-vector instance: dropped)
      const int64_t nblk = (int64_t)(c / u) * samples;
      const int64_t bytes = nvec * 16 * samples * (c / u);
      bool done = nblk >= 128 || bytes <= (4 << 20);
      if (!done) {}
      else if (fits(256, 4)) {
        if (silu) hipLaunchKernelGGL((gn_onepass_kernel<256, 4, true>), grid, dim3(256), 0, s, xb, yb, gamma, beta, rows, c, vu, eps, pf);
        else hipLaunchKernelGGL((gn_onepass_kernel<256, 4, false>), grid, dim3(256), 0, s, xb, yb, gamma, beta, rows, c, vu, eps, pf);
      } else if (fits(256, 13)) {
        if (silu) hipLaunchKernelGGL((gn_onepass_kernel<256, 13, true>), grid, dim3(256), 0, s, xb, yb, gamma, beta, rows, c, vu, eps, pf);
        else hipLaunchKernelGGL((gn_onepass_kernel<256, 13, false>), grid, dim3(256), 0, s, xb, yb, gamma, beta, rows, c, vu, eps, pf);
      }
      else done = false;
      // (round 6: a 768-thread / 17-vector instance -- slabs of up to 204 KiB in ONE block's registers: the per-frame norms
      // of level 0, 256 slabs = one per CU, and the clip-wide norms of level 2 -- was built, passed its tests and LOST:
      // 38 -> 60 us, 19.5 -> 31 us, -2.1 % per guided forward (profiles/r06_gn_onepass768_bench.txt,
      // r06_gn_onepass768_forward_ab.txt).  With one block per CU the chip reads, reduces and writes in lock-step -- no
      // block's stores overlap another's loads -- and a CU alone draws a fraction of its share of the fabric; the three
      // launches stream both passes at 4.4-5.5 TB/s.  Removed.)
      if (done) {
        TC_LAUNCH_CHECK();
        return TC_OK;
      }
    }
  }
  int nch, cr;
  gn_chunking(samples, rows, &nch, &cr);
  float* part = reinterpret_cast<float*>(workspace);
  float* stats = part + (int64_t)samples * nch * 64;
  dim3 grid(nch, samples), block(GN_THREADS);
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, s, reinterpret_cast<const bf16_t*>(x), part, rows, c, nch, cr);
  TC_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((samples * 32 + 3) / 4), block, 0, s, part, stats, samples, rows, c, nch,
                     eps);
  TC_LAUNCH_CHECK();
  // the weight-prefetch planes ride on the LAST of the three launches: the one right in front of the consumer
  const dim3 agrid(nch, samples, 1 + pf_planes(pf, (int64_t)nch * samples));
  if (silu) hipLaunchKernelGGL(gn_apply_kernel<true>, agrid, block, 0, s, reinterpret_cast<const bf16_t*>(x),
                               reinterpret_cast<bf16_t*>(y), gamma, beta, stats, rows, c, cr, pf);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, agrid, block, 0, s, reinterpret_cast<const bf16_t*>(x),
                          reinterpret_cast<bf16_t*>(y), gamma, beta, stats, rows, c, cr, pf);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_groupnorm(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                            int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                            void* workspace, int64_t workspace_bytes, void* stream) {
  return groupnorm_impl(x, y, gamma, beta, samples, rows, c, eps, silu, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int tc_groupnorm_pf(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                               int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                               void* workspace, int64_t workspace_bytes, const TcPrefetch* prefetch, void* stream) {
  return groupnorm_impl(x, y, gamma, beta, samples, rows, c, eps, silu, workspace, workspace_bytes, prefetch, stream);
}

extern "C" int tc_groupnorm_part(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta, const float* part,
                                 int32_t part_rows, int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !y || !gamma || !beta || !part || !workspace || samples <= 0 || rows <= 0 || c <= 0 || part_rows <= 0) return TC_EINVAL;
  if ((c % 32) != 0 || (c % 8) != 0 || c > GN_MAX_SLOTS * GN_THREADS * 8 || c > 4096 || samples > 65535) return TC_ESHAPE;
  if ((rows % part_rows) != 0) return TC_ESHAPE;                 // a block of partial sums may not straddle two samples
  if (!tc_aligned16(x) || !tc_aligned16(y)) return TC_EALIGN;
  if (workspace_bytes < tc_groupnorm_workspace(samples, rows, c)) return TC_EWORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int nch, cr;
  gn_chunking(samples, rows, &nch, &cr);
  float* stats = reinterpret_cast<float*>(workspace) + (int64_t)samples * nch * 64;
  hipLaunchKernelGGL(gn_finalize_part_kernel, dim3(samples * 32), dim3(256), 0, s, part, stats, samples, rows, c,
                     part_rows, eps);
  TC_LAUNCH_CHECK();
  if (silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nch, samples), dim3(GN_THREADS), 0, s, reinterpret_cast<const bf16_t*>(x),
                               reinterpret_cast<bf16_t*>(y), gamma, beta, stats, rows, c, cr, PfArgs{});
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nch, samples), dim3(GN_THREADS), 0, s, reinterpret_cast<const bf16_t*>(x),
                          reinterpret_cast<bf16_t*>(y), gamma, beta, stats, rows, c, cr, PfArgs{});
  TC_LAUNCH_CHECK();
  return TC_OK;
}

static int layernorm_impl(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                          int32_t rows, int32_t c, float eps, const TcPrefetch* prefetch, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || c <= 0) return TC_EINVAL;
  if ((c % 8) != 0 || c > 4 * 64 * 8) return TC_ESHAPE;
  if (!tc_aligned16(x) || !tc_aligned16(y) || !tc_aligned16(gamma) || !tc_aligned16(beta)) return TC_EALIGN;
  PfArgs pf;
  if (const int rc = pf_args(prefetch, &pf)) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  bf16_t* yb = reinterpret_cast<bf16_t*>(y);
  const int rpb = c <= 512 ? 16 : (c <= 1024 ? 8 : 4);       // rows per block
  const unsigned blocks = (unsigned)((rows + rpb - 1) / rpb);
  const dim3 grid(blocks, 1 + pf_planes(pf, blocks));
  if (c <= 512)
    hipLaunchKernelGGL((layernorm_kernel<1, 4>), grid, dim3(256), 0, st, xb, yb, gamma, beta, rows, c, eps, nullptr, 0, nullptr, 0, pf);
  else if (c <= 1024)
    hipLaunchKernelGGL((layernorm_kernel<2, 2>), grid, dim3(256), 0, st, xb, yb, gamma, beta, rows, c, eps, nullptr, 0, nullptr, 0, pf);
  else
    hipLaunchKernelGGL((layernorm_kernel<4, 1>), grid, dim3(256), 0, st, xb, yb, gamma, beta, rows, c, eps, nullptr, 0, nullptr, 0, pf);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_layernorm(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                            int32_t rows, int32_t c, float eps, void* stream) {
  return layernorm_impl(x, y, gamma, beta, rows, c, eps, nullptr, stream);
}

extern "C" int tc_layernorm_pf(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                               int32_t rows, int32_t c, float eps, const TcPrefetch* prefetch, void* stream) {
  return layernorm_impl(x, y, gamma, beta, rows, c, eps, prefetch, stream);
}

extern "C" int tc_layernorm_mxfp8(const tc_bf16* x, uint8_t* q, int32_t ldq, uint8_t* sc, int32_t lds, const float* gamma,
                                  const float* beta, int32_t rows, int32_t c, float eps, void* stream) {
  if (!x || !q || !sc || !gamma || !beta || rows <= 0 || c <= 0) return TC_EINVAL;
  if ((c % 32) != 0 || c > 4 * 64 * 8 || ldq < c || lds * 32 < c) return TC_ESHAPE;
  if (!tc_aligned16(x) || !tc_aligned16(gamma) || !tc_aligned16(beta) || (reinterpret_cast<uintptr_t>(q) & 7u) || (ldq & 7)) return TC_EALIGN;
  const int vpr = c >> 3, pad = lds - (c >> 5);
  const int nv = c <= 512 ? 1 : (c <= 1024 ? 2 : 4);
  if (vpr + pad > 64 * nv) return TC_ESHAPE;               // no idle lanes left to zero the scale padding
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  if (c <= 512)
    hipLaunchKernelGGL((layernorm_kernel<1, 4, true>), dim3((rows + 15) / 16), dim3(256), 0, st, xb, nullptr, gamma, beta, rows, c, eps, q, ldq, sc, lds);
  else if (c <= 1024)
    hipLaunchKernelGGL((layernorm_kernel<2, 2, true>), dim3((rows + 7) / 8), dim3(256), 0, st, xb, nullptr, gamma, beta, rows, c, eps, q, ldq, sc, lds);
  else
    hipLaunchKernelGGL((layernorm_kernel<4, 1, true>), dim3((rows + 3) / 4), dim3(256), 0, st, xb, nullptr, gamma, beta, rows, c, eps, q, ldq, sc, lds);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_softmax_rows(const float* s, tc_bf16* p, int32_t rows, int32_t n, int32_t n_out, int32_t lds,
                               int32_t ldo, int32_t causal_period, void* stream) {
  if (!s || !p || rows <= 0 || n <= 0 || n_out < n || lds < n || ldo < n_out || causal_period < 0) return TC_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), s,
                     reinterpret_cast<bf16_t*>(p), n, n_out, lds, ldo, causal_period);
  TC_LAUNCH_CHECK();
  return TC_OK;
}
