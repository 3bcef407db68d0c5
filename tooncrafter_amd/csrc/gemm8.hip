// 8-wave 256x256 ping-pong bf16 MFMA GEMM, persistent over output tiles, gfx950.
//
// Same contract as gemm.hip (implicit-GEMM gather, fused epilogue, LDS-DMA tile loads into an XOR-swizzled LDS
// image, XCD-aware tile order).  What differs is the MAIN LOOP and what happens BETWEEN tiles.
//
// Main loop.  The 4-wave kernels run every wave through the same sequence  wait -> barrier -> fragment reads -> MFMAs
// in lock-step: a K-step costs its LDS reads PLUS its MFMAs (profiles/r01_v6_gemm_ablation.txt: each is ~65 % of the
// loop and they only half overlap).  Here a block is 8 waves = two per SIMD (MI355X_MICROARCH.md "Two waves per
// SIMD"), split into two GROUPS of four (one wave per SIMD each) that run the same phase sequence ONE BARRIER
// INTERVAL APART:
//
//     interval      0      1      2      3      4   ...
//     group 0      R0     M0     R1     M1     R2          R = ds_read the fragments of the phase + request one half-tile
//     group 1       -     R0     M0     R1     M1          M = the phase's 8 MFMAs (256 matrix-pipe cycles)
//
// so in every interval each SIMD has one wave feeding its matrix pipe and one wave reading LDS / issuing LDS-DMA:
// matrix beside memory, never matrix beside matrix.  A fragment read is issued a whole interval before its first
// MFMA, so LDS latency never stands in front of the matrix pipe.
//
// Tile 256 x 256 x 64; wave (grp, wc) owns rows grp*128..+128, columns wc*64..+64 = 4 x 2 v_mfma_f32_32x32x16_bf16
// tiles (128 fp32 accumulators per lane).  A K-tile is four phases, one output quadrant (64 x 32) each:
//     P0: read A0 B0 -> Q(0,0)    P1: read B1 -> Q(0,1)    P2: read A1 -> Q(1,1)    P3: (B0 kept) -> Q(1,0)
// 24 ds_read_b128 per wave per K-tile for 32 MFMAs (0.75; the 128x128 kernel needs 1.0).
// LDS: 2 buffers x 4 HALF-TILES of 16 KiB, organised by quadrant use, not by tile row: A-half h holds tile rows
// {g*128 + h*64 + r}, B-half h the W rows {wc*64 + h*32 + r} -- every half-tile is read in ONE phase, so it can be
// re-requested while the rest of its K-tile is still being consumed.  One half-tile (2 LDS-DMA pieces per thread)
// is requested per phase, 1.5 K-tiles ahead of its first read:
//     P0: B1(t+1)   P1: A1(t+1)   P2: B0(t+2)   P3: A0(t+2); P0 / P1 / P3 each end with s_waitcnt vmcnt(8): the half-tile the
//     next phase reads has landed, the four requested after it stay in flight
// WAR: a half-tile is re-requested >= 2 phases after its last ds_read (the reader's lgkmcnt wait precedes a barrier the
// requester has passed, also across the group stagger).  RAW: the counted vmcnt of every wave precedes two barriers
// before the first read of K-tile t+1 (one for the other group's wait).  Never vmcnt(0) in the steady state, raw
// s_barrier only (a __syncthreads() would drain the DMA queue; cdna_hip_programming.md "Pipelining across barriers").
//
// Between tiles.  One block per CU (128 KiB of stage buffers) has no second block to hide a tile's first-load latency
// and its epilogue behind -- with K = 320 ... 1280 (5 ... 20 K-tiles) that is most of a short tile's life.  The block
// therefore WALKS a strided sequence of tiles, and the K-tile stream simply continues across the tile boundary: the
// requests of the last two K-tiles of tile i are the first 1.5 K-tiles of tile i+1 (the gather state switches to the
// next tile at P2 / P3 of K-tile nk-2), so the next tile's operands land while the accumulators of tile i go through the
// epilogue.  The epilogue uses the 32 KiB of LDS the stage buffers leave free (8 private 4 KiB slabs), contains no
// barrier, and -- because group 1 runs an interval behind -- starts under the other group's last MFMAs and ends under
// its first.  Per wave: eight passes of 16 rows through the slab, 16-byte row vectors (bias / row bias / activation /
// GEGLU / residual / store), the global operands of pass n+1 requested before pass n is computed.
#include "gemm_common.h"
#include "gemm_persist.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int G8_BM = 256, G8_BN = 256, G8_THREADS = 512;
constexpr int G8_HALF = 128 * TC_BK * 2;     // one half-tile: 128 rows x 128 B = 16 KiB
constexpr int G8_BUF = 4 * G8_HALF;          // A0 | A1 | B0 | B1
constexpr int G8_WN = 64;                    // columns per wave
constexpr int G8_SLAB = 16 * G8_WN * 4;      // per-wave epilogue slab: 16 rows x 64 fp32

// AB: compile-time ablations for scripts/gemm8_bench (TC_G8_ABLATE, linear gather only; results are WRONG with any bit
// but 8 / 16 / 32 set):  1 no MFMAs | 2 no tile requests after the prologue | 4 no fragment reads after the first K-tile
//   8 no s_setprio | 16 no group stagger (all eight waves in lock-step) | 32 fragment reads waited for BEFORE the barrier
//   64 no epilogue
template <int GATHER, int AB = 0>
__global__ __launch_bounds__(G8_THREADS, 2) void gemm8_kernel(const TcGemmParams p, const int total_tiles, const int stagger) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * G8_BUF + 8 * G8_SLAB];      // 160 KiB: the whole CU

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases travel in M0
  const int grp = wave_u >> 2;                                   // row half of the tile AND the phase stagger
  const int wc = wave_u & 3;

  const int tiles_n = (p.n + G8_BN - 1) / G8_BN;
  const int tiles_m = (p.m + G8_BM - 1) / G8_BM;
  const int64_t bz = blockIdx.z;
  const g8_srd_t w_rsrc = g8_make_srd(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of smem
  const bool k_ragged = (p.k & (TC_BK - 1)) != 0;
  const int nk = (p.k + TC_BK - 1) / TC_BK;                       // >= 2 (host)

  // ---- loader geometry.  One wave instruction fills 1 KiB = 8 rows of a half-tile, lane l at (row l>>3, physical
  // chunk l&7): the XOR swizzle is applied to the SOURCE chunk.  A half-tile is two passes of 64 rows; thread ->
  // LDS row lrow + 64 i.  A-half h, LDS row j  <->  tile row (j>>6)*128 + h*64 + (j&63) = lrow + 64 (2 i + h);
  // B-half h, LDS row j  <->  W row (j>>5)*64 + h*32 + (j&31).
  const int lrow = tid >> 3;                                      // 0..63
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  // the REQUEST state: gather offsets of the tile whose K-tiles are being requested (the current tile until P2 of its
  // K-tile nk-2, the next one from then on)
  G8Gather<GATHER> ag;
  g8_srd_t a_rsrc;
  uint32_t b_voff[2][2];
  // K-tile -> tap without a division per request: tap = kt / (cin / 64) as a multiply-high (exact for kt < 1024 and
  // cin <= 4096: the host checks)
  const int tpt = GATHER == TC_GATHER_LINEAR ? 1 : p.cin / TC_BK;
  const uint32_t tap_magic = (65536u + tpt - 1) / tpt;
  auto set_tile_b = [&](int id, int& tm, int& tn) {      // the cheap half of the request state: the W rows
    g8_tile_of(id, tiles_m, tiles_n, total_tiles, tm, tn);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = tn * G8_BN + ((lrow >> 5) + 2 * i) * 64 + h * 32 + (lrow & 31);
        b_voff[h][i] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
      }
  };
  auto set_tile_a = [&](int tm) {                         // ... and the A rows (divisions: once per tile)
    ag.init(p, tm * G8_BM, lrow, chunk);
    a_rsrc = g8_make_srd(reinterpret_cast<const bf16_t*>(p.a) + bz * p.stride_a + ag.row_lo * p.lda, tc_a_extent(p) - ag.row_lo * p.lda * 2);
  };

  auto stage_a = [&](auto H_, int kt, int boff) {           // boff: byte offset of the LDS buffer (0 or G8_BUF), uniform
    constexpr int h = decltype(H_)::value;
    const int k0 = kt * TC_BK;
    int tap = 0;
    uint32_t soff = (uint32_t)k0 * 2u, delta = 0;
    if (GATHER != TC_GATHER_LINEAR) {
      tap = (int)(((uint32_t)kt * tap_magic) >> 16);
      soff = (uint32_t)(k0 - tap * p.cin) * 2u;
      if (GATHER == TC_GATHER_CONV3x3) {
        const int ty = (tap * 11) >> 5;                   // tap / 3 for tap < 9
        delta = (uint32_t)(((ty - 1) * p.w_in + (tap - ty * 3 - 1)) * p.lda * 2);
      } else {
        delta = (uint32_t)((tap - 1) * p.h_out * p.w_out * p.lda * 2);
      }
    }
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    const uint32_t dst = lds0 + boff + h * G8_HALF + wave_u * 1024;
    g8_dma16(a_rsrc, dst, ag.voff(h, tap, delta) | kill, soff);
    g8_dma16(a_rsrc, dst + 8192, ag.voff(2 + h, tap, delta) | kill, soff);
  };
  auto stage_b = [&](auto H_, int kt, int boff) {
    constexpr int h = decltype(H_)::value;
    const int k0 = kt * TC_BK;
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    const uint32_t dst = lds0 + boff + (2 + h) * G8_HALF + wave_u * 1024;
    g8_dma16(w_rsrc, dst, b_voff[h][0] | kill, (uint32_t)k0 * 2u);
    g8_dma16(w_rsrc, dst + 8192, b_voff[h][1] | kill, (uint32_t)k0 * 2u);
  };

  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // ---- fragment geometry (v_mfma_f32_32x32x16_bf16): lane holds row lane&31 of its 32-row tile, k = 8 (lane>>5) .. +7
  // of the 16-deep slice kk -> one ds_read_b128 at logical chunk 2 kk + (lane>>5).  The rows a wave reads start at
  // multiples of 32, so the swizzle term (row>>1)&7 only depends on the lane.
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int sw = (frow >> 1) & 7;
  int coff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) coff[kk] = ((kk * 2 + fhalf) ^ sw) << 4;
  const int a_row_off = (grp * 64 + frow) * (TC_BK * 2);
  const int b_row_off = (wc * 32 + frow) * (TC_BK * 2);

  bf16x8 fa[2][4], fb0[4], fb1[4];
  bool abl_started = false;                      // ablation 4 only: fragments are read once, by the first K-tile
  if constexpr ((AB & 4) != 0) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fa[1][kk] = fb0[kk] = fb1[kk] = bf16x8{}; }
  }
  auto read_a = [&](auto H_, int boff) {
    constexpr int h = decltype(H_)::value;
    if constexpr ((AB & 4) != 0) { if (abl_started) return; }
    const char* base = smem + boff + h * G8_HALF + a_row_off;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fa[ib][kk] = *reinterpret_cast<const bf16x8*>(base + ib * 4096 + coff[kk]);
  };
  auto read_b = [&](auto H_, int boff, bf16x8 (&fb)[4]) {
    constexpr int h = decltype(H_)::value;
    if constexpr ((AB & 4) != 0) { if (abl_started) return; }
    const char* base = smem + boff + (2 + h) * G8_HALF + b_row_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fb[kk] = *reinterpret_cast<const bf16x8*>(base + coff[kk]);
  };
  auto mma = [&](auto MH_, auto NH_, bf16x8 (&fb)[4]) {
    constexpr int mh = decltype(MH_)::value, nh = decltype(NH_)::value;
    if constexpr ((AB & 1) != 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(fa[0][kk]), "v"(fa[1][kk]), "v"(fb[kk]));
      return;
    }
    if constexpr ((AB & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
        acc[mh * 2 + ib][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ib][kk], fb[kk], acc[mh * 2 + ib][nh], 0, 0, 0);
    if constexpr ((AB & 8) == 0) __builtin_amdgcn_s_setprio(0);
  };
  auto r_end = [&]() {                           // end of an R segment
    if constexpr ((AB & 32) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_barrier();
  };

  // ---- tile walk: block b takes tiles b, b + G, b + 2G, ...  (G a multiple of 8: every tile of a block keeps the
  // block's XCD in g8_tile_of's deal)
  int tile_id = blockIdx.x;
  int cur_m, cur_n, nxt_m = 0, nxt_n = 0;
  set_tile_b(tile_id, cur_m, cur_n);
  set_tile_a(cur_m);
  bool has_next = tile_id + (int)gridDim.x < total_tiles;

  // one K-tile = four phases, read from the LDS buffer at byte offset `bo` (a scalar: the fragment addresses are eight
  // v_add per K-tile, the DMA destinations travel in M0 anyway).  K-tile t of the current tile requests K-tiles t+1
  // (P0, P1) and t+2 (P2, P3) of the K-tile STREAM, which runs on into the next tile of this block
  auto ktile = [&](int bo, int t) {
    const int nbo = bo ^ G8_BUF;
    const bool n1 = (AB & 2) ? false : (t + 1 < nk || has_next), n2 = (AB & 2) ? false : (t + 2 < nk || has_next);
    const int k1 = t + 1 < nk ? t + 1 : 0;
    const int k2 = t + 2 < nk ? t + 2 : t + 2 - nk;
    // Waits: a half-tile is waited for in the phase BEFORE its first read (the other group's wait needs one more
    // barrier), by a count that leaves the FOUR half-tiles requested after it in flight (8 pieces per thread = 64 KiB
    // per CU): every request gets at least four phases -- a whole K-tile of MFMA time -- to land.  (The first version
    // waited once per K-tile with two half-tiles left in flight; B1 / A1 then had two phases, and the no-MFMA ablation
    // ran at the latency of the DMA, not at its bandwidth.)  `n1` also says whether the PREVIOUS K-tile issued its
    // P2 / P3 requests (the same condition one K-tile later).
    // P0: reads A0 B0 (waited for at the previous P3); requests B1(t+1); waits for B1(t)
    read_a(ic<0>{}, bo);
    read_b(ic<0>{}, bo, fb0);
    if (n1) {
      stage_b(ic<1>{}, k1, nbo);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // after B1(t): A1(t) B0(t+1) A0(t+1) B1(t+1)
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // after B1(t): A1(t) only
    }
    r_end();
    mma(ic<0>{}, ic<0>{}, fb0);
    g8_barrier();
    // P1: reads B1; requests A1(t+1); waits for A1(t)
    read_b(ic<1>{}, bo, fb1);
    if (n1) {
      stage_a(ic<1>{}, k1, nbo);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // after A1(t): B0(t+1) A0(t+1) B1(t+1) A1(t+1)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    r_end();
    mma(ic<0>{}, ic<1>{}, fb1);
    g8_barrier();
    // P2: reads A1; requests B0(t+2).  From K-tile nk-2 on, the request state belongs to the next tile of this block
    read_a(ic<1>{}, bo);
    if (n2) {
      if (t + 2 == nk) set_tile_b(tile_id + gridDim.x, nxt_m, nxt_n);
      stage_b(ic<0>{}, k2, bo);
    }
    r_end();
    mma(ic<1>{}, ic<1>{}, fb1);
    g8_barrier();
    // P3: (no fragment reads: the fewest live registers of the K-tile -- the A rows of the next tile are set up here);
    // requests A0(t+2); waits for B0(t+1) A0(t+1)
    if (n2) {
      if (t + 2 == nk) set_tile_a(nxt_m);
      stage_a(ic<0>{}, k2, bo);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // after A0(t+1): B1(t+1) A1(t+1) B0(t+2) A0(t+2)
    } else if (n1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // after A0(t+1): B1(t+1) A1(t+1)
    }
    r_end();
    mma(ic<1>{}, ic<0>{}, fb0);
    g8_barrier();
  };

  // ---- epilogue of the tile (tm, tn): per wave, 8 passes of 16 rows through a private fp32 slab.  No barrier; the
  // global operands (row bias, residual) of pass n+1 are requested before pass n is computed.
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  float* slab = reinterpret_cast<float*>(smem + 2 * G8_BUF) + wave_u * (G8_SLAB / 4);
  const bf16_t* res_base = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + bz * p.stride_c : nullptr;
  char* c_base = reinterpret_cast<char*>(p.c) + bz * p.stride_c * 2;

  struct PassOps {            // global operands of one pass of a NON-GEGLU epilogue: 2 row vectors per lane
    u32x4 res[2];
    f32x4 rb[2][2];
  };
  auto epilogue = [&](int tm, int tn) {
    const int col_w0 = tn * G8_BN + wc * G8_WN;          // first packed column of this wave
    const int row_w0 = tm * G8_BM + grp * 128;
    auto spill = [&](auto I_, auto H_) {                 // accumulators of pass (i, half) -> slab
      constexpr int i = decltype(I_)::value, half = decltype(H_)::value;
      // accumulator registers r = 8*half .. 8*half+7 hold local rows (r&3) + 4*fhalf + 8*((r>>2)&1)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * half + q;
          const int lr = (r & 3) + 4 * fhalf + 8 * ((r >> 2) & 1);
          slab[lr * G8_WN + j * 32 + frow] = acc[i][j][r];
        }
    };
    if (!geglu) {
      // lane -> vector column vc = lane & 7 (8 columns), slab rows lane>>3 and 8 + (lane>>3): the lane's 8 output
      // columns are the same in every pass, so the bias is loaded once
      const int vc = lane & 7, lr0 = lane >> 3;
      const int n0 = col_w0 + vc * 8;
      const bool col_ok = n0 < p.n;
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.bias && col_ok) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bv[e]));       // (same reason as in finish() below)
      auto fetch = [&](int pass, PassOps& o) {
        const int row_base = row_w0 + pass * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = row_base + lr0 + 8 * q;
          const int mc = m < p.m ? m : p.m - 1;
          o.res[q] = u32x4{0u, 0u, 0u, 0u};
          o.rb[q][0] = o.rb[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (col_ok) {
            if (res_base) o.res[q] = *reinterpret_cast<const u32x4*>(res_base + (int64_t)mc * p.ldr + n0);
            if (p.row_bias) {
              const float* rp = p.row_bias + (int64_t)(mc / p.row_div) * p.ldrb + n0;
              o.rb[q][0] = *reinterpret_cast<const f32x4*>(rp);
              o.rb[q][1] = *reinterpret_cast<const f32x4*>(rp + 4);
            }
          }
        }
      };
      auto finish = [&](int pass, const PassOps& o) {
        const int row_base = row_w0 + pass * 16;
        // every fetched register is "used" here on every path: hipcc sinks the arithmetic below into the store's
        // branch, and a load whose only use was skipped would stay pending in its scoreboard -- it then protects the
        // register with s_waitcnt vmcnt(0) in front of the K loop's fragment reads, which drains the DMA stream
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" ::"v"(o.res[q]), "v"(o.rb[q][0]), "v"(o.rb[q][1]));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int lr = lr0 + 8 * q;
          const int m = row_base + lr;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + vc * 8);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + vc * 8 + 4);
          float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float rbv = e < 4 ? o.rb[q][0][e] : o.rb[q][1][e - 4];
            x[e] = (x[e] + bv[e]) + rbv;                 // the 4-wave kernels' order of additions: bit-identical results
          }
          if (res_base) {
            float rf[8];
            unpack8(o.res[q], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += rf[e];
          }
          if (m < p.m && col_ok)
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = pack8(x);
        }
      };
      // (unrolled: a rolled loop makes hipcc index the accumulators dynamically, i.e. through scratch.  The passes are
      // kept SMALL instead -- with every epilogue option compiled in they were 60 KiB of code, more than the instruction
      // cache, and streaming them cost ~20 us per tile)
      PassOps o0, o1;
      fetch(0, o0);
      spill(ic<0>{}, ic<0>{}); fetch(1, o1); finish(0, o0);
      spill(ic<0>{}, ic<1>{}); fetch(2, o0); finish(1, o1);
      spill(ic<1>{}, ic<0>{}); fetch(3, o1); finish(2, o0);
      spill(ic<1>{}, ic<1>{}); fetch(4, o0); finish(3, o1);
      spill(ic<2>{}, ic<0>{}); fetch(5, o1); finish(4, o0);
      spill(ic<2>{}, ic<1>{}); fetch(6, o0); finish(5, o1);
      spill(ic<3>{}, ic<0>{}); fetch(7, o1); finish(6, o0);
      spill(ic<3>{}, ic<1>{}); finish(7, o1);
    } else {
      // packed columns: every 32 = [16 values | 16 gates]; output vector u (8 columns) of a slab row reads values
      // at packed 32*(u/2) + 8*(u&1) and gates 16 further; 16 rows x 4 vectors = one per lane.  GEGLU launches carry
      // no row bias and no residual
      const int lr = lane >> 2, u = lane & 3;
      const int pc = 32 * (u >> 1) + 8 * (u & 1);
      const int n0 = (col_w0 >> 1) + u * 8;
      const bool col_ok = n0 < n_out;
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.bias && col_ok) {
        const float* bp = p.bias + col_w0 + pc;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(bp + 16), g1 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; bg[e] = g0[e]; bg[4 + e] = g1[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bv[e]), "v"(bg[e]));   // loads consumed on every path (see above)
      auto finish = [&](int pass) {
        const int m = row_w0 + pass * 16 + lr;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + pc);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + pc + 4);
        const f32x4 glo = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + pc + 16);
        const f32x4 ghi = *reinterpret_cast<const f32x4*>(slab + lr * G8_WN + pc + 20);
        float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const float gt[8] = {glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {        // pairs: packed fp32 arithmetic (common.h gelu_erf_f2)
          const tc_f32x2 v = {x[e] + bv[e], x[e + 1] + bv[e + 1]};
          const tc_f32x2 h = v * gelu_erf_f2(tc_f32x2{gt[e] + bg[e], gt[e + 1] + bg[e + 1]});
          x[e] = h[0]; x[e + 1] = h[1];
        }
        if (m < p.m && col_ok)
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(c_base) + (int64_t)m * p.ldc + n0) = pack8(x);
      };
      spill(ic<0>{}, ic<0>{}); finish(0);
      spill(ic<0>{}, ic<1>{}); finish(1);
      spill(ic<1>{}, ic<0>{}); finish(2);
      spill(ic<1>{}, ic<1>{}); finish(3);
      spill(ic<2>{}, ic<0>{}); finish(4);
      spill(ic<2>{}, ic<1>{}); finish(5);
      spill(ic<3>{}, ic<0>{}); finish(6);
      spill(ic<3>{}, ic<1>{}); finish(7);
    }
  };

  // De-phasing (TC_G8_STAGGER, units of 64 * 127 cycles): every other CU of an XCD starts late, so that the blocks of a
  // launch do not all reach their epilogues -- a burst of 128 KiB of stores per CU -- at the same moment
  if (stagger > 0 && ((blockIdx.x >> 3) & 1)) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue: K-tile 0 whole, the first halves of K-tile 1
  stage_a(ic<0>{}, 0, 0);
  stage_b(ic<0>{}, 0, 0);
  stage_b(ic<1>{}, 0, 0);
  stage_a(ic<1>{}, 0, 0);
  stage_b(ic<0>{}, 1, G8_BUF);
  stage_a(ic<0>{}, 1, G8_BUF);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // A0(0) B0(0) have landed; B1(0) A1(0) B0(1) A0(1) in flight
  g8_barrier();
  if constexpr ((AB & 16) == 0) { if (grp == 1) g8_barrier(); }   // the stagger: group 1 runs one barrier interval behind group 0

  int bo = 0;                                    // LDS buffer (byte offset) of the stream's current K-tile
  for (;;) {
    zero_acc();
    for (int t = 0; t < nk; ++t) {
      ktile(bo, t);
      bo ^= G8_BUF;
      if constexpr ((AB & 4) != 0) abl_started = true;
    }
    if constexpr ((AB & 64) != 0) {
      float tsum = 0.f;
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) tsum += acc[i][j][r];
      if (tsum == 1.2345e30f) reinterpret_cast<float*>(p.c)[tid] = tsum;
    } else {
      epilogue(cur_m, cur_n);
    }
    if (!has_next) break;
    tile_id += gridDim.x;
    cur_m = nxt_m; cur_n = nxt_n;
    has_next = tile_id + (int)gridDim.x < total_tiles;
  }
  if constexpr ((AB & 16) == 0) { if (grp == 0) g8_barrier(); }   // realign: every wave has executed the same number of barriers
}

int g8_mode() {        // TC_GEMM8 = 0 never | 1 heuristic (default) | 2 whenever the shape allows; read per call (A/B runs)
  const char* e = getenv("TC_GEMM8");
  return e ? atoi(e) : 1;
}

}  // namespace

// Decide whether the 8-wave kernel should take this (already validated) GEMM, and launch it.  1 = launched.
int tc_gemm8_try(const TcGemmParams& p, int batch, hipStream_t s, bool dry) {
  const int mode = g8_mode();
  if (mode == 0) return 0;
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  if ((n_out & 7) != 0 || (geglu && (p.n & 31) != 0)) return 0;   // vector epilogue only; GEGLU packs per 32
  if (p.k <= TC_BK) return 0;                                     // the K-tile stream needs two K-tiles per tile
  // the epilogue is the common case only (plain or GEGLU, bf16 out): every option it carried was code the wave streams
  // through once per tile
  if ((p.act != TC_ACT_NONE && !geglu) || p.alpha != 1.f || p.out_scale != 1.f || p.out_f32) return 0;
  if (p.gather == TC_GATHER_CONV3x3 && (p.stride != 1 || p.upsample || p.pad != 1)) return 0;   // plain 3x3 only
  if (p.gather != TC_GATHER_LINEAR && (p.k / TC_BK >= 1024 || p.cin > 4096)) return 0;          // tap = multiply-high
  const int tiles_n = (p.n + G8_BN - 1) / G8_BN;
  const int tiles_m = (p.m + G8_BM - 1) / G8_BM;
  if (mode == 1) {
    // Measured (profiles/r04_gemm8_bench.txt, interleaved against the default routing): the kernel is ahead where ONE
    // round of 256x256 tiles fills the chip and K is long -- the level-1 ff2 (20480 x 640 x 2560: 240 tiles,
    // 1.11-1.21x), square problems from 4096 (1.00-1.07x), the decoder's 512-channel convolutions (1.03x) -- and
    // behind wherever the tile grid quantises badly on 256 CUs (every UNet width is 320 k, every row count 5 * 2^n:
    // N = 320 pads to 512, 320 / 160 / 100 tiles run 1.25 / 0.63 / 0.39 rounds) or K is short (K <= 1280: the epilogue
    // of a 256 x 256 tile -- 128 KiB of stores, for GEGLU as many VALU cycles as the tile's MFMAs -- is not overlapped
    // with matrix work when a CU holds a single block).  The heuristic takes the first class only.
    const int64_t tiles = (int64_t)tiles_n * tiles_m * batch;
    const double n_eff = (double)p.n / ((double)tiles_n * G8_BN);
    // (one round: linear layers only -- the convolutions of that size are on the 160x160-tile kernel, which the 8-wave
    // kernel does NOT beat: level-1 3x3 640 -> 640 0.85x, 1920 -> 640 0.80x)
    const bool one_round = tiles >= 224 && tiles <= 256 && p.gather == TC_GATHER_LINEAR;
    const bool many = tiles >= 1024 && p.n % G8_BN == 0 && p.n >= 512 && p.k >= 4096;   // (256 -> 256 convolutions: 0.89-0.92x)
    // GEGLU projections of levels 0 and 1 (81920 x 2560 x 320, 20480 x 5120 x 640: 3200 / 1600 tiles): once the GELU of
    // the epilogue lost its division sequence (common.h gelu_erf_f) the kernel is ahead there too -- 1.09-1.13x and
    // 1.06-1.09x (profiles/r04_gelu_ab.txt); level 2 (5120 x 10240 x 1280) stays behind (0.95x)
    const bool gated = geglu && p.gather == TC_GATHER_LINEAR && tiles >= 1536 && p.k <= 640 && p.n % G8_BN == 0;
    if (!gated && (geglu || p.k < 2048 || n_eff < 0.8 || !(one_round || many))) return 0;
  }
  const int64_t total = (int64_t)tiles_n * tiles_m;
  if (total > 0x7fffffffLL) return 0;
  if (dry) return 1;
  static const int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int gmax = [&] { const char* e = getenv("TC_G8_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : cus; }() & ~7;
  const int g = (int)(total < gmax ? total : gmax);         // a multiple of 8 unless it covers every tile in one round
  dim3 grid((unsigned)g, 1, (unsigned)batch), block(G8_THREADS);
  const int tt = (int)total;
  const int stg = [] { const char* e = getenv("TC_G8_STAGGER"); return e ? atoi(e) : 0; }();
#ifdef TC_TIMING_BUILDS      /* timing ablations: WRONG results by construction, never in the product library */
  const int ab = [] { const char* e = getenv("TC_G8_ABLATE"); return e ? atoi(e) : 0; }();
  if (ab && p.gather == TC_GATHER_LINEAR) {
#define TC_G8_AB(X) case X: hipLaunchKernelGGL((gemm8_kernel<TC_GATHER_LINEAR, X>), grid, block, 0, s, p, tt, stg); return 1
    switch (ab) {
      TC_G8_AB(1); TC_G8_AB(2); TC_G8_AB(3); TC_G8_AB(4); TC_G8_AB(6); TC_G8_AB(7); TC_G8_AB(8); TC_G8_AB(16); TC_G8_AB(32); TC_G8_AB(64); TC_G8_AB(65);
      default: break;
    }
#undef TC_G8_AB
  }
#endif
  switch (p.gather) {
    case TC_GATHER_LINEAR: hipLaunchKernelGGL((gemm8_kernel<TC_GATHER_LINEAR>), grid, block, 0, s, p, tt, stg); break;
    case TC_GATHER_CONV3x3: hipLaunchKernelGGL((gemm8_kernel<TC_GATHER_CONV3x3>), grid, block, 0, s, p, tt, stg); break;
    default: hipLaunchKernelGGL((gemm8_kernel<TC_GATHER_CONVT3>), grid, block, 0, s, p, tt, stg); break;
  }
  return 1;
}
