// Helpers shared by the GEMM kernels (gemm.hip: 128x128 / 64x64 tiles, gemm_wide.hip: 256-row tiles).
#pragma once
#include "common.h"

#include <stdlib.h>

constexpr int TC_BK = 64;   // K-step of every GEMM kernel: one 128-byte LDS row per tile row

// Byte offset of 16-byte chunk `chunk` (0..7) of tile row `row` in the LDS image.  Rows are 128 B; the
// chunk index is XOR-swizzled by (row>>1)&7, which makes the MFMA fragment reads (16 lanes reading
// the same logical chunk of 16 different rows with ds_read_b128) bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * (TC_BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == TC_ACT_SILU) return silu_f(v);
  if (act == TC_ACT_GELU) return gelu_erf_f(v);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Tile loads go through buffer_load_dwordx4 (raw buffer, 128-bit SRD in SGPRs):
//     address = base(SRD) + voffset(VGPR, per lane) + soffset(SGPR, per K-step)
// * the per-lane voffset (row * lda + chunk) is computed ONCE per block; advancing along K is a scalar
//   add on soffset -- the first version of this kernel recomputed 64-bit row addresses per load and
//   spent 12-15 VALU instructions per MFMA doing so (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA), which,
//   not memory latency, was what held it at ~25 % of the MFMA roof;
// * rows that do not exist (M/N tails, convolution zero padding, K tails) get voffset = TC_OOB, which
//   is >= num_records, so the hardware returns zeros: no masking instructions at all.
// Offsets are 31-bit, so the descriptor of the A operand does not start at the tensor but at the lowest source row the
// BLOCK's tile can touch (tc_tile_row_lo: a 64-bit, block-uniform base folded into the SRD); per-lane offsets are
// relative to that row.  A tile's rows span a few image rows (3x3), 2 frames + the tile (temporal taps) or the tile itself
// (linear), so an activation of ANY size is addressable -- the batched decode of BASELINE configs[3] (2.7 GB per
// level-0 tensor at two 320x512 clips) goes through one launch; the host checks the per-block span, not the tensor.
// The descriptor's num_records is the TRUE byte extent of the operand (last row + its K columns), so a
// mis-computed row / tap offset that leaves the operand reads zeros instead of whatever VA follows the
// allocation (or faulting where nothing is mapped there); TC_OOB is >= any extent by construction.
constexpr uint32_t TC_OOB = 0x80000000u;
constexpr int TC_SRD_FLAGS = 0x00020000;

typedef __amdgpu_buffer_rsrc_t tc_rsrc_t;

__device__ __forceinline__ tc_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  const int rec = bytes < 0x7ffffff0LL ? (int)bytes : 0x7ffffff0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, rec, TC_SRD_FLAGS);
}

// byte extents of one batch item of A (source rows of the gather) and W
__host__ __device__ __forceinline__ int64_t tc_a_rows(const TcGemmParams& p) {
  return p.gather == TC_GATHER_CONV3x3 ? (int64_t)p.frames * p.h_in * p.w_in : (int64_t)p.m;
}
__host__ __device__ __forceinline__ int64_t tc_a_extent(const TcGemmParams& p) {
  const int kc = p.gather == TC_GATHER_LINEAR ? p.k : p.cin;
  return ((tc_a_rows(p) - 1) * p.lda + kc) * 2;
}
__host__ __device__ __forceinline__ int64_t tc_w_extent(const TcGemmParams& p) {
  return ((int64_t)(p.n - 1) * p.ldw + p.k) * 2;
}

__device__ __forceinline__ u32x4 buf_load16(tc_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// Direct global -> LDS load of one 1-KiB piece per wave instruction (buffer_load_dwordx4 ... lds):
// lane l's 16 bytes land at lds + 16 l.  `lds` must be wave-uniform (it travels in M0).  Out-of-range
// offsets write zeros, like the register form.  Kept in a __device__ function: the host pass of hipcc
// silently drops a __global__ template that names this builtin directly.
__device__ __forceinline__ void glds16(tc_rsrc_t rsrc, char* lds, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// Block -> output tile.  Blocks are dealt round-robin to the 8 XCDs (blockIdx & 7), each with its own L2, so
// M-tile t always lives on XCD t & 7 and its A rows are fetched into one L2 only.  Within an XCD:
//   order 0: all N-tiles of an M-tile back to back (A stays hot; W is re-streamed once per M-tile -- fine
//            while the weight matrix fits the 4 MiB L2);
//   order g > 0: N-tiles in chunks of g (8): for each chunk, all M-tiles of the XCD, so ~64 co-resident blocks
//            cover 8 M-tiles x 8 N-tiles and both operands are re-read from L2, not from the fabric (for the
//            wide-N layers whose W does not fit L2).
// (Giving each XCD a CONTIGUOUS range of M-tiles instead -- so that 3x3 / temporal convolutions re-read their halo
// rows from one L2 -- was measured 0.7 % slower on the UNet than this interleaved deal, and dropped.)
__device__ __forceinline__ void tc_tile_of_block(int bid, int tiles_m, int tiles_n, int order, int& tile_m, int& tile_n) {
  const int xcd = bid & 7;
  const int slot = bid >> 3;
  if (order < 0) {
    // W-stationary walk for the low-resolution layers (few rows, big weight matrices): tiles in N-major order, cut
    // into 8 contiguous runs, one per XCD -- an XCD touches ~tiles_n / 8 + 1 N-tiles of W instead of all of them
    // (order 0 makes every L2 fetch the whole weight matrix and, with tiles_m = 10, gives two XCDs twice the work)
    const int total = tiles_m * tiles_n;
    const int per = (total + 7) >> 3;
    const int lin = xcd * per + slot;
    if (slot >= per || lin >= total) { tile_m = tiles_m; tile_n = 0; return; }   // surplus block: caller exits
    tile_n = lin / tiles_m;
    tile_m = lin - tile_n * tiles_m;
    return;
  }
  if (order == 0) {
    tile_m = (slot / tiles_n) * 8 + xcd;
    tile_n = slot % tiles_n;
    return;
  }
  const int GN = order;                                // N-tiles per chunk
  const int tm_x = (tiles_m + 7) >> 3;                 // M-tiles per XCD (upper bound; surplus blocks exit)
  const int full = (tiles_n / GN) * GN;
  int m_local;
  if (slot < tm_x * full) {
    const int c = slot / (tm_x * GN), r = slot - c * (tm_x * GN);
    m_local = r / GN;
    tile_n = c * GN + (r - m_local * GN);
  } else {
    const int rem = tiles_n - full, r = slot - tm_x * full;
    m_local = r / rem;
    tile_n = full + (r - m_local * rem);
  }
  tile_m = m_local * 8 + xcd;
}

// host side: order 1 when the weight matrix outgrows one XCD's L2 and there are enough N-tiles to chunk
inline int tc_gemm_tile_order(const TcGemmParams& p, int tiles_n) {
  // TC_GEMM_ORDER = chunk width in N-tiles (default 8; 0 = always the plain walk); TC_GEMM_ORDER_MIB = weight
  // size from which the chunked walk is used (default 4 = one XCD's L2)
  static const int chunk = [] { const char* e = getenv("TC_GEMM_ORDER"); return e ? atoi(e) : 8; }();
  static const int64_t min_bytes = [] { const char* e = getenv("TC_GEMM_ORDER_MIB");
                                        return (int64_t)((e ? atof(e) : 4.0) * (1 << 20)); }();
  return (chunk > 0 && tiles_n >= 2 * chunk && (int64_t)p.n * p.ldw * 2 > min_bytes) ? chunk : 0;
}

// Lowest source row of A that the tile whose first output row is `tile_row0` can touch (block-uniform; a lower bound).
// Rows of one tile ascend in (frame, y, x), and so do their source rows, so the first row's top-left tap bounds them all.
template <int GATHER>
__device__ __forceinline__ int64_t tc_tile_row_lo(const TcGemmParams& p, int tile_row0) {
  if (GATHER == TC_GATHER_LINEAR) return tile_row0;
  if (GATHER == TC_GATHER_CONVT3) {
    const int64_t r = (int64_t)tile_row0 - (int64_t)p.h_out * p.w_out;
    return r > 0 ? r : 0;
  }
  const int q = tile_row0 / p.w_out;
  const int f = q / p.h_out, y = q - f * p.h_out;
  int iy = y * p.stride - p.pad;
  if (iy < 0) iy = 0;
  if (p.upsample) iy >>= 1;
  const int64_t r = ((int64_t)f * p.h_in + iy) * p.w_in - 1;
  return r > 0 ? r : 0;
}

// SRD of the A operand for one block: base = first byte of row `row_lo` of batch item bz, records = what is left of the
// operand's true extent from there (clamped to the 31-bit range by make_rsrc)
__device__ __forceinline__ tc_rsrc_t tc_a_rsrc(const TcGemmParams& p, int64_t bz, int64_t row_lo) {
  return make_rsrc(reinterpret_cast<const bf16_t*>(p.a) + bz * p.stride_a + row_lo * p.lda, tc_a_extent(p) - row_lo * p.lda * 2);
}

// Per-thread gather state for `R` rows of the A tile.
template <int GATHER, int R>
struct AGather {
  uint32_t base[R];    // byte offset of (row, chunk) from row_lo -- the centre tap for convolutions; TC_OOB if row >= M
  int64_t row_lo;      // tc_tile_row_lo of this block: the row the SRD of A starts at
  uint32_t vbits[R];   // convolution: bit t set if tap t of this row is inside the image
  int f[R], y[R], x[R];  // generic 3x3 path (stride 2 / fused upsample) only
  bool ok[R];
  // The row offsets of a convolution change with the TAP, i.e. every cin / 64 K-steps; between taps only the scalar offset
  // moves.  offsets() therefore keeps the offsets of the tap it last formed them for (5 masked adds per row and K-step,
  // and the K-step -> tap division, were 45 of the ~115 vector and most of the ~110 scalar instructions a wave spent per
  // K-step beside its 50 MFMAs: profiles/r04_pmc_g16.txt) and finds the tap by a multiply-high where that is exact.
  uint32_t cvoff[R];
  int ctap;              // the tap cvoff belongs to; -1: none yet
  uint32_t tap_magic;    // ceil(65536 / (cin / 64)): K-tile -> tap, exact for K-tiles < 1024 and cin <= 4096; 0: divide

  __device__ __forceinline__ bool fast3x3(const TcGemmParams& p) const {
    return p.stride == 1 && !p.upsample && p.pad == 1;
  }

  __device__ __forceinline__ void init(const TcGemmParams& p, int tile_row0, int lrow, int row_step, int chunk) {
    const int hw = p.h_out * p.w_out;
    row_lo = tc_tile_row_lo<GATHER>(p, tile_row0);
    ctap = -1;
    tap_magic = 0;
    if (GATHER != TC_GATHER_LINEAR && (p.cin % TC_BK) == 0 && p.cin <= 4096 && p.k < 1024 * TC_BK) {
      const uint32_t tpt = (uint32_t)(p.cin / TC_BK);
      tap_magic = (65536u + tpt - 1) / tpt;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int mm = tile_row0 + lrow + row_step * i;
      ok[i] = mm < p.m;
      const int mc = ok[i] ? mm : 0;
      f[i] = y[i] = x[i] = 0;
      vbits[i] = ok[i] ? 0xffffffffu : 0u;
      if (GATHER == TC_GATHER_LINEAR) {
        base[i] = ok[i] ? (uint32_t)(((int64_t)mc - row_lo) * p.lda * 2 + chunk * 16) : TC_OOB;
      } else if (GATHER == TC_GATHER_CONV3x3) {
        const int q = mc / p.w_out;
        x[i] = mc - q * p.w_out;
        f[i] = q / p.h_out;
        y[i] = q - f[i] * p.h_out;
        // (rows past M take mc = 0, whose offset may wrap: every tap of theirs is masked by vbits / ok)
        base[i] = (uint32_t)(((((int64_t)f[i] * p.h_in + y[i]) * p.w_in + x[i]) - row_lo) * p.lda * 2 + chunk * 16);
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y[i] + t / 3 - 1, ix = x[i] + t % 3 - 1;
          if (ok[i] && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) bits |= 1u << t;
        }
        vbits[i] = bits;
      } else {  // CONVT3
        base[i] = (uint32_t)(((int64_t)mc - row_lo) * p.lda * 2 + chunk * 16);
        const int tt = (mc / hw) % p.t_len;
        vbits[i] = ok[i] ? ((tt > 0 ? 1u : 0u) | 2u | (tt + 1 < p.t_len ? 4u : 0u)) : 0u;
      }
    }
  }

  // voffset of every row and the scalar soffset for the K-block starting at k0 (a multiple of TC_BK)
  __device__ __forceinline__ void offsets(const TcGemmParams& p, int k0, int chunk, uint32_t (&voff)[R], uint32_t& soff) {
    if (GATHER == TC_GATHER_LINEAR) {
      soff = (uint32_t)k0 * 2u;
#pragma unroll
      for (int i = 0; i < R; ++i) voff[i] = base[i];
      return;
    }
    const int tap = tap_magic ? (int)((((uint32_t)k0 / TC_BK) * tap_magic) >> 16) : k0 / p.cin;
    soff = (uint32_t)(k0 - tap * p.cin) * 2u;
    if (tap != ctap) {                                   // block-uniform
      ctap = tap;
      if (GATHER == TC_GATHER_CONV3x3) {
        const int dy = tap / 3 - p.pad, dx = tap - (tap / 3) * 3 - p.pad;
        if (fast3x3(p)) {
          const uint32_t delta = (uint32_t)((dy * p.w_in + dx) * p.lda * 2);
#pragma unroll
          for (int i = 0; i < R; ++i) cvoff[i] = ((vbits[i] >> tap) & 1u) ? base[i] + delta : TC_OOB;
        } else {
          const int hv = p.upsample ? p.h_in * 2 : p.h_in;
          const int wv = p.upsample ? p.w_in * 2 : p.w_in;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            int iy = y[i] * p.stride + dy;
            int ix = x[i] * p.stride + dx;
            const bool v = ok[i] && iy >= 0 && iy < hv && ix >= 0 && ix < wv;
            if (p.upsample) { iy >>= 1; ix >>= 1; }
            const int64_t src = ((int64_t)f[i] * p.h_in + iy) * p.w_in + ix;
            cvoff[i] = v ? (uint32_t)((src - row_lo) * p.lda * 2 + chunk * 16) : TC_OOB;
          }
        }
      } else {  // CONVT3
        const uint32_t delta = (uint32_t)((tap - 1) * p.h_out * p.w_out * p.lda * 2);
#pragma unroll
        for (int i = 0; i < R; ++i) cvoff[i] = ((vbits[i] >> tap) & 1u) ? base[i] + delta : TC_OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) voff[i] = cvoff[i];
  }
};

// Host-side guard shared by the launchers: every byte offset a tile load can form must fit the 31-bit range the
// out-of-range marker relies on.  For A that is the span of ONE block's source rows from its tc_tile_row_lo (tiles of
// at most 256 output rows), not the tensor.
inline bool tc_gemm_offsets_fit(const TcGemmParams& p) {
  int64_t span_rows = 256;
  if (p.gather == TC_GATHER_CONVT3) span_rows += 2 * (int64_t)p.h_out * p.w_out;
  else if (p.gather == TC_GATHER_CONV3x3)
    span_rows = ((int64_t)256 / (p.w_out > 0 ? p.w_out : 1) + 2) * p.stride * p.w_in + 4 * (int64_t)p.w_in + 8;
  const int64_t a_bytes = (span_rows + 1) * p.lda * 2;
  const int64_t w_bytes = (int64_t)p.n * p.ldw * 2;
  return a_bytes < 0x7fffff00LL && w_bytes < 0x7fffff00LL;
}

// host side: -1 (W-stationary walk) for the lowest-resolution layers (M <= 2048 rows) whose weight matrix outweighs
// their activation rows.  Measured (profiles/r02_nmajor_ab.txt): +5-7 % on the level-3 convolutions and qkv, neutral
// on the rest of level 3, 3-7 % SLOWER on the level-2 convolutions (40 M-tiles: the A halo re-reads cost more than
// the W re-reads the Infinity Cache was already absorbing) -- hence the row limit.
// TC_GEMM_NMAJOR = 0 never, 1 heuristic (default), 2 always; read per call.
inline bool tc_gemm_nmajor(const TcGemmParams& p) {
  const char* e = getenv("TC_GEMM_NMAJOR");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0 || (p.batch > 1)) return false;
  if (mode == 2) return true;
  return p.m <= 2048 && tc_w_extent(p) > tc_a_extent(p);
}

// (dry = true: the routing decision only, nothing is launched -- tc_gemm_gn_rows asks every family in tc_gemm_bf16's order)
int tc_gemm_wide_try(const TcGemmParams& p, int batch, hipStream_t s, bool force, bool dry = false);   // gemm_wide.hip; 1 = launched
int tc_gemm_tile16_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry = false);   // gemm16.hip; 1 = launched
int tc_gemm_ws_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry = false);       // gemm_ws.hip (K = 320); 1 = launched
int tc_gemm8_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry = false);         // gemm8.hip (8-wave 256x256 ping-pong); 1 = launched
int tc_conv_halo_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry = false);     // conv_halo.hip (tap-reuse patches; TC_CONV_HALO: 1 = the measured routing (default), 0 = never, 2 = strict); 1 = launched, -1 = strict mode declined
