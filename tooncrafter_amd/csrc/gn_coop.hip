// GroupNorm(32) (+SiLU) as ONE launch that reads x ONCE and writes y once (ABI 11: tc_groupnorm_coop) -- the algorithmic
// 2 B + 2 B per element, where the three-launch path of norm.hip moves 2 + 2 + 2 (x is read by the statistics pass and again
// by the apply pass) and pays three dependent launches.  Replaces GroupNormSpecific / nn.GroupNorm + SiLU of
// lvdm/basics.py:76-87, openaimodel3d.py:152-154,176-179,255-266, attention.py:254,340 for the tensors whose (sample, unit)
// slabs do not fit ONE block (gn_onepass_kernel's case): UNet level 0 (52-157 MB), the clip-wide norms of levels 1 / 2.
//
// How: the chip's register files hold the tensor.  A persistent grid of at most `capacity` co-resident 512-thread blocks
// (2 per CU at <= 128 VGPRs: 512 blocks x 512 threads x 16 vectors x 16 B = 64 MB) splits every sample's rows into `nch` chunks;
// a block keeps its chunk in registers (packed bf16, NV 16-byte vectors per thread, every thread pinned to one 8-channel
// column so per-channel sums stay in registers and rows are read as whole 2C-byte lines), writes the chunk's 32 (sum, sum of
// squares) pairs, and meets the other blocks of ITS SAMPLE at a counter: the last one to arrive reduces the sample's partials
// in fp64 in chunk order (so the result does not depend on who was last: bit-reproducible) and publishes (mean, rstd); every
// block then normalises from its registers.  Tensors above the capacity run as `rounds` of whole samples through the same
// blocks (the per-frame norms of the 640 / 960-channel concatenations at level 0).
//
// The inter-block exchange uses NO fences.  norm.hip records what an agent-scope release per block costs on this chip (the
// L2 write-back it implies: +27 % on a clip); here every shared word is a single-location ATOMIC at agent scope (relaxed):
// partials and statistics travel as 64-bit exchanges (returning: `s_waitcnt vmcnt(0)` then proves they were performed at the
// coherence point before the counter moves), counters as fetch-adds, readers use atomic loads -- the memory model keeps such
// accesses coherent by themselves, and ordinary loads of x / stores of y never need ordering against them.
// Deadlock freedom: the host launches at most `capacity` blocks (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs, queried
// once per instance), so all blocks of a round are resident; a spin is bounded anyway (GC_SPIN_LIMIT polls, then the block
// poisons its statistics with NaN instead of hanging the GPU).
// Counters: `sync` is int32 [samples][4] = (arrived, ready, departed, -), ZERO before the first call; the last block to leave a
// sample zeroes its three words again, so the buffer is zero between launches (stream order) and is never touched by the host.
#include "common.h"
#include "gn_route.h"

#include <stdlib.h>

namespace {

constexpr int GC_T = 512;
constexpr int GC_NV_MAX = 16;
constexpr int GC_SPIN_LIMIT = 1 << 21;          // x ~0.3 us per poll: ~0.6 s, three orders above any legitimate wait

typedef unsigned long long u64_t;
typedef __amdgpu_buffer_rsrc_t tc_rsrc_t;
constexpr uint32_t GC_OOB = 0x80000000u;      // >= any chunk extent (a chunk is at most 16 x 8 KiB)
constexpr int GC_SRD_FLAGS = 0x00020000;       // raw buffer, as gemm_common.h

__device__ __forceinline__ uint32_t gc_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t gc_load64(const u64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t gc_xchg64(u64_t* p, u64_t v) {
  return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gc_xchg(uint32_t* p, uint32_t v) {
  return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gc_add(uint32_t* p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t gc_pack(float lo, float hi) {
  return (u64_t)__float_as_uint(lo) | ((u64_t)__float_as_uint(hi) << 32);
}
// every returning atomic this wave has issued has been performed (and nothing below moves above: compiler barrier)
// the packed words become "new" values: the compiler may not keep their unpacked fp32 copies alive from the statistics pass
// to the apply pass (8 registers per vector instead of 4)
__device__ __forceinline__ void gc_opaque(u32x4& v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }
__device__ __forceinline__ void gc_performed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NV, bool SILU>
__global__ __launch_bounds__(GC_T, 4) void gn_coop_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          u64_t* part, u64_t* stats, uint32_t* sync, int samples, int rows, int c,
                                                          int nch, int chunk_rows, int spr, int rounds, float eps) {
  __shared__ float red[GC_T * 16];            // [q][row lane][channel]: rows_pp * c <= 512 * 8 per q
  __shared__ double gpd[2][32];
  __shared__ float gp[2][32];
  __shared__ float st[64];
  __shared__ int flag, spin_ok;
  const int tid = threadIdx.x;
  const int vpr = c >> 3, cpg = c >> 5;
  const int rows_pp = GC_T / vpr;
  const int rlane = tid / vpr, col = tid - rlane * vpr;
  const bool live = rlane < rows_pp;
  const int sub = blockIdx.x / nch, chunk = blockIdx.x - sub * nch;
  const int r0 = chunk * chunk_rows;
  const int r1 = min(rows, r0 + chunk_rows);
  // the thread's channels never change: first group and the position inside it (groups of the other 7 follow by counting)
  const int g_first = live ? (col * 8) / cpg : 0, g_rem = live ? col * 8 - g_first * cpg : 0;
  // the fold's role of this thread: quantity q, group fg, quarter fl
  const int fq = tid >> 7, fg = (tid >> 2) & 31, fl = tid & 3;

  for (int round = 0; round < rounds; ++round) {
    const int sample = round * spr + sub;
    if (sample >= samples) break;                                     // uniform over the block
    // addresses = the chunk's buffer descriptor (scalar registers; its extent is the chunk, so anything outside reads zeros
    // / is not written) + i * (rows_pp rows) as the scalar offset + ONE 32-bit per-thread offset; a pass the thread has no
    // row in gets the out-of-range offset: no branches, no 64-bit per-row addresses
    const int64_t cbytes = (int64_t)(r1 - r0) * c * 2;
    const tc_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x + ((int64_t)sample * rows + r0) * c), 0, (int)cbytes, GC_SRD_FLAGS);
    const tc_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + ((int64_t)sample * rows + r0) * c, 0, (int)cbytes, GC_SRD_FLAGS);
    const uint32_t toff = (uint32_t)(rlane * c + col * 8) * 2u;
    const int nrow = live ? r1 - r0 - rlane : 0;                        // this thread's rows: i * rows_pp < nrow
    const uint32_t pass = (uint32_t)(rows_pp * c) * 2u;
    u32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, i * rows_pp < nrow ? toff : GC_OOB, i * pass, 0));
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float f[8];
      unpack8(v[i], f);                                               // rows beyond the chunk hold zeros: they add nothing
#pragma unroll
      for (int e = 0; e < 8; ++e) { sum[e] += f[e]; sq[e] = __builtin_fmaf(f[e], f[e], sq[e]); }
      // one vector at a time: without this the scheduler unpacks ALL NV vectors first (8 NV live fp32 values: spills at NV = 16)
      asm volatile("" : "+v"(sum[0]), "+v"(sum[1]), "+v"(sum[2]), "+v"(sum[3]), "+v"(sum[4]), "+v"(sum[5]), "+v"(sum[6]), "+v"(sum[7]),
                        "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]), "+v"(sq[4]), "+v"(sq[5]), "+v"(sq[6]), "+v"(sq[7]));
    }
    if (live) {
      float* d0 = red + (int64_t)rlane * c + col * 8;
      float* d1 = red + (int64_t)(rows_pp + rlane) * c + col * 8;
      *reinterpret_cast<f32x4*>(d0) = f32x4{sum[0], sum[1], sum[2], sum[3]};
      *reinterpret_cast<f32x4*>(d0 + 4) = f32x4{sum[4], sum[5], sum[6], sum[7]};
      *reinterpret_cast<f32x4*>(d1) = f32x4{sq[0], sq[1], sq[2], sq[3]};
      *reinterpret_cast<f32x4*>(d1 + 4) = f32x4{sq[4], sq[5], sq[6], sq[7]};
    }
    __syncthreads();
    if (tid < 256) {                                                  // channels x row lanes -> groups, fixed order
      float a = 0.f;
      const float* base = red + (int64_t)fq * rows_pp * c + fg * cpg;
#pragma unroll 1
      for (int rl = 0; rl < rows_pp; ++rl) {
#pragma unroll 2
        for (int ch = fl; ch < cpg; ch += 4) a += base[rl * c + ch];
      }
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      if (fl == 0) gp[fq][fg] = a;
    }
    __syncthreads();
    // ---- publish the chunk's partials, arrive
    uint32_t* sy = sync + (int64_t)sample * 4;
    if (tid < 64) {
      // the exchange's RETURN is what proves it was performed: keep the result alive so that it stays a returning atomic
      u64_t old = 0;
      if (tid < 32) old = gc_xchg64(part + ((int64_t)sample * nch + chunk) * 32 + tid, gc_pack(gp[0][tid], gp[1][tid]));
      gc_performed();
      if (old == 0x7ff8dead7ff8beefULL) gp[0][0] = 0.f;               // never true for a pair of finite sums
      if (tid == 0) flag = gc_add(sy, 1u) == (uint32_t)(nch - 1);
    }
    __syncthreads();
    if (flag) {                                                       // last to arrive: every partial of the sample is in memory
      if (tid < 256) {
        double a = 0.0;
        const uint32_t* pp = reinterpret_cast<const uint32_t*>(part + (int64_t)sample * nch * 32) + fg * 2 + fq;
#pragma unroll 4
        for (int k = fl; k < nch; k += 4) a += (double)__uint_as_float(gc_load(pp + (int64_t)k * 64));
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (fl == 0) gpd[fq][fg] = a;
      }
      __syncthreads();
      if (tid < 64) {
        u64_t old2 = 0;
        if (tid < 32) {
          const double cnt = (double)rows * cpg;
          const double mean = gpd[0][tid] / cnt;
          double var = gpd[1][tid] / cnt - mean * mean;
          if (var < 0.0) var = 0.0;
          old2 = gc_xchg64(stats + (int64_t)sample * 32 + tid, gc_pack((float)mean, (float)(1.0 / sqrt(var + (double)eps))));
        }
        gc_performed();
        if (old2 == 0x7ff8dead7ff8beefULL) gp[0][0] = 0.f;
        if (tid == 0) (void)gc_xchg(sy + 1, 1u);                      // ready
      }
    }
    if (tid == 0) {
      int n = 0;
#pragma unroll 1
      while (gc_load(sy + 1) == 0u && ++n < GC_SPIN_LIMIT) __builtin_amdgcn_s_sleep(8);
      spin_ok = n < GC_SPIN_LIMIT;
    }
    __syncthreads();
    if (tid < 32) {
      const u64_t ms = gc_load64(stats + (int64_t)sample * 32 + tid);
      const bool ok = spin_ok != 0;
      st[tid * 2] = ok ? __uint_as_float((uint32_t)ms) : __builtin_nanf("");
      st[tid * 2 + 1] = ok ? __uint_as_float((uint32_t)(ms >> 32)) : __builtin_nanf("");
    }
    __syncthreads();
    if (tid == 0 && gc_add(sy + 2, 1u) == (uint32_t)(nch - 1)) {     // last to leave: every block of the sample has its statistics
      __hip_atomic_store(sy, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sy + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sy + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (live) {
      float sc[8], sh[8];
      {
        const f32x4 ga0 = *reinterpret_cast<const f32x4*>(gamma + col * 8), ga1 = *reinterpret_cast<const f32x4*>(gamma + col * 8 + 4);
        const f32x4 be0 = *reinterpret_cast<const f32x4*>(beta + col * 8), be1 = *reinterpret_cast<const f32x4*>(beta + col * 8 + 4);
        int g = g_first, r = g_rem;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = st[g * 2 + 1] * (e < 4 ? ga0[e] : ga1[e - 4]);
          sc[e] = a;
          sh[e] = (e < 4 ? be0[e] : be1[e - 4]) - st[g * 2] * a;
          if (++r == cpg) { r = 0; ++g; }
        }
      }
      uint32_t chain = 0;                                              // orders the passes: see below
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        // one vector at a time (as in the statistics pass): the next vector's words are "produced" after this one's result
        asm volatile("" : "+v"(v[i][0]), "+v"(v[i][1]), "+v"(v[i][2]), "+v"(v[i][3]) : "v"(chain));
        if (i * rows_pp < nrow) {
          float f[8];
          unpack8(v[i], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = __builtin_fmaf(f[e], sc[e], sh[e]);
            f[e] = SILU ? silu_f(t) : t;
          }
          const u32x4 o = pack8(f);
          chain = o[3];
          __builtin_amdgcn_raw_buffer_store_b128(o, yr, toff, i * pass, 0);
        }
      }
    }
    if (rounds > 1) __syncthreads();                                  // `st`, `red`, `gp`, `flag` are rewritten by the next round
  }
}

struct GcPlan {
  int nch, chunk_rows, spr, rounds, nv, grid;
};

template <int NV, bool SILU>
int gc_capacity_of() {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, gn_coop_kernel<NV, SILU>, GC_T, 0) != hipSuccess) return 0;
  return cus * per;
}

// co-resident blocks the chip gives EVERY instance (the plan does not know which one it will launch), at most 1024 (the
// workspace layout's chunks per sample); TC_GN_COOP_CAP (read per call: the tests force many rounds with it) lowers it
int gc_capacity() {
  static const int hw = [] {
    int m = gc_capacity_of<16, true>();
    const int o[] = {gc_capacity_of<16, false>(), gc_capacity_of<8, true>(), gc_capacity_of<8, false>(),
                     gc_capacity_of<4, true>(), gc_capacity_of<4, false>()};
    for (int v : o) m = v < m ? v : m;
    return m < 0 ? 0 : (m > 1024 ? 1024 : m);
  }();
  const char* e = getenv("TC_GN_COOP_CAP");
  const int lim = e ? atoi(e) : 0;
  return lim > 0 && lim < hw ? lim : hw;
}

}  // namespace

// The decomposition for (samples, rows, c) on `capacity` co-resident blocks; false = this kernel does not take the problem.
// Host-callable without a device (tests/test_gn_coop_cpu.py checks the invariants the kernel relies on).
static bool gc_plan(int samples, int rows, int c, int capacity, GcPlan* p) {
  if (samples <= 0 || rows <= 0 || c <= 0 || (c % 32) != 0 || c > 4096 || capacity <= 0 || samples > 65535) return false;
  const int vpr = c >> 3;
  const int rows_pp = GC_T / vpr;                                        // >= 1: c <= 4096
  const int64_t max_rows = (int64_t)GC_NV_MAX * rows_pp;                  // rows one block can hold
  const int64_t nch_min = (rows + max_rows - 1) / max_rows;
  if (nch_min > capacity) return false;                                 // one sample does not fit the chip's registers
  const int spr_max = (int)(capacity / nch_min);
  const int rounds = (samples + spr_max - 1) / spr_max;
  const int spr = (samples + rounds - 1) / rounds;
  int64_t nch = capacity / spr;                                         // as many chunks as stay co-resident ...
  const int64_t useful = (rows + 2 * rows_pp - 1) / (2 * rows_pp);       // ... but at least two row passes per block
  if (nch > useful) nch = useful;
  if (nch < nch_min) nch = nch_min;
  const int chunk_rows = (int)((rows + nch - 1) / nch);
  nch = (rows + chunk_rows - 1) / chunk_rows;
  const int need = (chunk_rows + rows_pp - 1) / rows_pp;
  p->nv = need <= 4 ? 4 : (need <= 8 ? 8 : 16);
  if (need > GC_NV_MAX) return false;
  p->nch = (int)nch;
  p->chunk_rows = chunk_rows;
  p->spr = spr;
  p->rounds = rounds;
  p->grid = spr * (int)nch;
  return p->grid <= capacity;
}

// for the CPU tests: the plan as six ints (nch, chunk_rows, spr, rounds, nv, grid); 0 = not taken
extern "C" int tc_groupnorm_coop_plan(int32_t samples, int32_t rows, int32_t c, int32_t capacity, int32_t* out6) {
  GcPlan p;
  if (!out6 || !gc_plan(samples, rows, c, capacity, &p)) return 0;
  out6[0] = p.nch; out6[1] = p.chunk_rows; out6[2] = p.spr; out6[3] = p.rounds; out6[4] = p.nv; out6[5] = p.grid;
  return 1;
}

extern "C" int64_t tc_groupnorm_coop_workspace(int32_t samples, int32_t rows, int32_t c) {
  (void)rows; (void)c;
  if (samples <= 0) return 0;
  // partials: at most `capacity` blocks per round, i.e. <= 1024 chunks per sample whatever the plan; statistics: 32 pairs per sample
  return ((int64_t)samples * 1024 * 32 + (int64_t)samples * 32) * 8;
}

extern "C" int64_t tc_groupnorm_coop_sync_bytes(int32_t samples) { return samples > 0 ? (int64_t)samples * 16 : 0; }

// > 0: the grid tc_groupnorm_coop would launch; 0: it does not take this problem (caller keeps tc_groupnorm).
// TC_GN_COOP (read per call: A/B runs) = 0 never | 1 / unset: where the single-block one-pass kernel does not apply (it keeps
// the per-frame norms of levels 1-3, whose slabs fit a block and need no exchange) | 2: wherever a plan exists
extern "C" int tc_groupnorm_coop_grid(int32_t samples, int32_t rows, int32_t c) {
  const char* e = getenv("TC_GN_COOP");
  const int mode = e ? atoi(e) : 1;
  if (mode <= 0) return 0;
  if (mode == 1) {
    static const bool onepass = [] { const char* o = getenv("TC_GN_ONEPASS"); return !(o && o[0] == '0'); }();
    if (onepass && gn_onepass_rule(samples, rows, c).nv) return 0;
  }
  GcPlan p;
  return gc_plan(samples, rows, c, gc_capacity(), &p) ? p.grid : 0;
}

extern "C" int tc_groupnorm_coop(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta, int32_t samples,
                                 int32_t rows, int32_t c, float eps, int32_t silu, void* workspace, int64_t workspace_bytes,
                                 void* sync, int64_t sync_bytes, void* stream) {
  if (!x || !y || !gamma || !beta || !workspace || !sync || samples <= 0 || rows <= 0 || c <= 0) return TC_EINVAL;
  if (!tc_aligned16(x) || !tc_aligned16(y) || !tc_aligned16(gamma) || !tc_aligned16(beta) || !tc_aligned16(workspace) ||
      (reinterpret_cast<uintptr_t>(sync) & 3u)) return TC_EALIGN;
  if (workspace_bytes < tc_groupnorm_coop_workspace(samples, rows, c)) return TC_EWORKSPACE;
  if (sync_bytes < tc_groupnorm_coop_sync_bytes(samples)) return TC_EWORKSPACE;
  GcPlan p;
  if (!gc_plan(samples, rows, c, gc_capacity(), &p)) return TC_ESHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  u64_t* part = reinterpret_cast<u64_t*>(workspace);
  u64_t* stats = part + (int64_t)samples * 1024 * 32;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  bf16_t* yb = reinterpret_cast<bf16_t*>(y);
  uint32_t* sy = reinterpret_cast<uint32_t*>(sync);
#define TC_GC_LAUNCH(NV_)                                                                                               \
  do {                                                                                                                  \
    if (silu) hipLaunchKernelGGL((gn_coop_kernel<NV_, true>), dim3(p.grid), dim3(GC_T), 0, s, xb, yb, gamma, beta, part, stats, \
                                 sy, samples, rows, c, p.nch, p.chunk_rows, p.spr, p.rounds, eps);                      \
    else hipLaunchKernelGGL((gn_coop_kernel<NV_, false>), dim3(p.grid), dim3(GC_T), 0, s, xb, yb, gamma, beta, part, stats,    \
                            sy, samples, rows, c, p.nch, p.chunk_rows, p.spr, p.rounds, eps);                           \
  } while (0)
  if (p.nv == 4) TC_GC_LAUNCH(4);
  else if (p.nv == 8) TC_GC_LAUNCH(8);
  else TC_GC_LAUNCH(16);
#undef TC_GC_LAUNCH
  TC_LAUNCH_CHECK();
  return TC_OK;
}
