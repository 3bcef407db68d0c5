// Pieces shared by the persistent 8-wave GEMM kernels (gemm8.hip: 256x256 ping-pong; gemm_ap.hip: two 128x128 groups in
// anti-phase): the XCD-aware bijective tile walk, the slim per-tile gather state, LDS-DMA requests from inline asm.
#pragma once
#include "gemm_common.h"

#include <type_traits>

namespace {

template <int N>
using ic = std::integral_constant<int, N>;

// Tile id -> (tile_m, tile_n), a bijection on [0, tiles_m * tiles_n).  Blocks are dealt round-robin to the 8 XCDs
// (id & 7), each with its own L2: XCD x walks a CONTIGUOUS range of the linear tile order below, and that order runs
// N-major through groups of 4 M-tiles, so the ~32 tiles an XCD has in flight (one per CU) form a 4 x 8 patch of the
// output: per K-tile they fetch 4 A half... 12 operand panels instead of the 33 a row of 32 tiles would.
constexpr int G8_GM = 4;
__device__ __forceinline__ void g8_tile_of(int id, int tiles_m, int tiles_n, int total, int& tm, int& tn) {
  const int x = id & 7, j = id >> 3;
  const int q = total >> 3, r = total & 7;
  const int lin = x * q + (x < r ? x : r) + j;
  const int per_group = G8_GM * tiles_n;
  const int g = lin / per_group, within = lin - g * per_group;
  const int gm = min(G8_GM, tiles_m - g * G8_GM);        // the last group may be short
  tn = within / gm;
  tm = g * G8_GM + (within - tn * gm);
}

// Request state of one tile's A rows: what AGather (gemm_common.h) keeps, minus the generic 3x3 path (stride 2 /
// fused upsample stay on the 4-wave kernels): per row a byte offset from the tile's lowest source row and a mask of
// the taps that fall inside the image.  Rows are lrow + STEP q, q = 0..R-1 (gemm8: 4 x 64; gemm16's interleaved loop: 5 x 32).
template <int GATHER, int STEP = 64, int R = 4>
struct G8Gather {
  uint32_t base[R];
  uint32_t vbits[R];
  int64_t row_lo;

  __device__ __forceinline__ void init(const TcGemmParams& p, int tile_row0, int lrow, int chunk) {
    row_lo = tc_tile_row_lo<GATHER>(p, tile_row0);
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int mm = tile_row0 + lrow + STEP * q;
      const bool ok = mm < p.m;
      const int mc = ok ? mm : 0;
      if (GATHER == TC_GATHER_LINEAR) {
        base[q] = ok ? (uint32_t)(((int64_t)mc - row_lo) * p.lda * 2 + chunk * 16) : TC_OOB;
        vbits[q] = 0;
      } else if (GATHER == TC_GATHER_CONV3x3) {
        const int qq = mc / p.w_out;
        const int x = mc - qq * p.w_out;
        const int f = qq / p.h_out;
        const int y = qq - f * p.h_out;
        base[q] = (uint32_t)(((((int64_t)f * p.h_in + y) * p.w_in + x) - row_lo) * p.lda * 2 + chunk * 16);
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
          if (ok && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) bits |= 1u << t;
        }
        vbits[q] = bits;
      } else {
        base[q] = (uint32_t)(((int64_t)mc - row_lo) * p.lda * 2 + chunk * 16);
        const int tt = (mc / (p.h_out * p.w_out)) % p.t_len;
        vbits[q] = ok ? ((tt > 0 ? 1u : 0u) | 2u | (tt + 1 < p.t_len ? 4u : 0u)) : 0u;
      }
    }
  }
  // byte offset of row q for the K-tile whose tap is `tap`; delta = the tap's row displacement in bytes (block-uniform)
  __device__ __forceinline__ uint32_t voff(int q, int tap, uint32_t delta) const {
    if (GATHER == TC_GATHER_LINEAR) return base[q];
    return ((vbits[q] >> tap) & 1u) ? base[q] + delta : TC_OOB;
  }
};

// LDS-DMA request issued from inline asm: hipcc then knows nothing about it -- it neither counts it in its own
// s_waitcnt vmcnt bookkeeping nor treats it as an LDS store.  With the builtin form (glds16) and a RUN-TIME buffer
// offset the compiler cannot tell the DMA destinations from the fragment reads of the other buffer and puts
// s_waitcnt vmcnt(0) in front of every fragment read (measured in the ISA); with compile-time buffers the K-tile body
// exists twice, the stream's parity has to be threaded through the tile boundaries and the register allocator spills.
// All ordering of these requests is therefore by hand (counted vmcnt + barriers, see the header); waits hipcc emits
// for its own loads can only be stricter than it thinks.  M0 (the LDS base) is written and restored inside the
// statement; s_nop 4: an SGPR written by SALU right before the statement may be read as descriptor / soffset.
typedef uint32_t g8_srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g8_srd_t g8_make_srd(const void* base, int64_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t rec = bytes < 0 ? 0u : (bytes < 0x7ffffff0LL ? (uint32_t)bytes : 0x7ffffff0u);
  g8_srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(rec);
  r[3] = TC_SRD_FLAGS;
  return r;
}
// a descriptor that lives across branches may end up in VGPRs (hipcc then cannot feed it to an "s" operand): re-assert
// that it is wave-uniform right in front of the requests that use it
__device__ __forceinline__ g8_srd_t g8_uniform(g8_srd_t v) {
  g8_srd_t r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_readfirstlane(v[i]);
  return r;
}
__device__ __forceinline__ void g8_dma16(g8_srd_t srd, uint32_t lds_dst, uint32_t voff, uint32_t soff) {
  uint32_t keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff)
      : "memory");
}

// Two requests in one statement: one hazard pad and one save / restore of M0 for the pair (9 instructions instead of 12;
// gemm16's interleaved loops issue their ten pieces as five pairs).
__device__ __forceinline__ void g8_dma16x2(g8_srd_t srd0, uint32_t dst0, uint32_t voff0, uint32_t soff0,
                                           g8_srd_t srd1, uint32_t dst1, uint32_t voff1, uint32_t soff1) {
  uint32_t keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %6, %7, %8 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(dst0), "v"(voff0), "s"(srd0), "s"(soff0), "s"(dst1), "v"(voff1), "s"(srd1), "s"(soff1)
      : "memory");
}

__device__ __forceinline__ void g8_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace
