// bf16 MFMA GEMM with implicit-GEMM row gather and fused epilogue, gfx950.
//
//   C[M, N] = epilogue( gather(A)[M, K] * W[N, K]^T )
//
// One kernel serves nn.Linear, Conv2d 1x1 / 3x3 (stride 1|2, optional fused nearest-x2
// upsample of the source) and Conv3d (3,1,1): activations are channels-last rows, so a
// convolution tap is just a different source row for the same K-slice of channels.
//
// Tiling (wave64): block tile 128x128x64, 4 waves as 2x2, each wave 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 sub-tiles (64 fp32 accumulators / lane).  A and B tiles travel
// global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB = 8 tile rows per wave
// instruction; no staging registers, no ds_write pass) into a double-buffered LDS image whose
// 16-byte chunks are XOR-swizzled by (row>>1)&7 -- applied on the source side of the DMA --
// which makes every ds_read_b128 lane group of the MFMA fragment reads conflict-free
// (MI355X_MICROARCH.md, LDS table).  The loads of tile k+1 are issued before the MFMAs of
// tile k, one vmcnt wait + barrier per K-step.  All tile loads are branch-free: out-of-range
// rows / taps / K-tails carry an out-of-range buffer offset and the hardware writes zeros.
// Measured (profiles/r01_v6_gemm_ablation.txt): replacing the register-staged loads by the
// DMA form is worth +9 % on the UNet; with it the K loop is bound by LDS/L1 traffic per FLOP
// of the 128x128 tile (loads alone and MFMAs alone each take ~65 % of the full loop).
// The epilogue transposes the accumulators through LDS (fp32) so the bias / embedding /
// activation / residual / GEGLU math and the global stores run on 16-byte row vectors, with
// a fully unrolled, unconditional fast path for whole vectors.  Blocks are numbered so that
// all N-tiles of one M-tile land on the same XCD (its L2 then serves the A re-reads).
#include "gemm_common.h"
#include "gemm_epilogue.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int BK = TC_BK;
// Tile = (64*TM) x (64*TN): 4 waves as 2x2, each wave (32*TM) x (32*TN) = TM x TN MFMA sub-tiles.
// 128x128 (TM=TN=2) is the general kernel; 64x64 (TM=TN=1) quadruples the block count for the
// low-resolution layers whose 128-tiles would not fill the 256 CUs; the big-M layers go to the
// 256-row kernel of gemm_wide.hip.

// PIPE: the K loop keeps TWO K-steps of tile loads in flight (both LDS stages requested before the first wait, stage k
// re-requested as soon as every wave has consumed it) with counted s_waitcnt vmcnt + raw s_barrier -- a __syncthreads()
// would drain the LDS-DMA queue (cdna_hip_programming.md, "Pipelining across barriers").  The plain loop has ONE K-step
// in flight behind vmcnt(0) + barrier: with K = 320-1280 a tile's life is mostly exposed load latency.
// (A per-wave epilogue -- private swizzled slabs, no block barrier, GEGLU evaluated in the accumulator layout after
// v_permlane16_swap so that only the product crosses LDS -- was built, bit-identical in 12 epilogue cases, and measured
// 1.01-1.02x on the GEGLU layers and 0.84-0.99x on the plain ones (128-byte instead of 256-byte row segments per store
// instruction), profiles/r03_wave_epilogue_ab.txt: the LDS transpose is not what the epilogue waits for.  Removed.)
// (A persistent form -- a block walks a strided sequence of tiles and requests the next tile's first K-step under the
// current tile's epilogue: two-pass transpose through one stage, static buffer-store count, counted vmcnt across the
// tile boundary -- was built, bit-identical in 13 cases, and measured 0.90-1.04x on every short-K shape,
// profiles/r03_persistent_gemm_ab.txt: the exposed first-load latency is not what the tile waits for either.  Removed.)
// (Measured on top of this loop and NOT kept, profiles/r03_*: starting the second block of each CU half a tile late
// (0.73-1.02x: co-resident blocks are not in lock-step), and two output tiles per block back to back so that the first
// tile's store acknowledgements arrive under the second K loop (0.90-1.09x where the grid stays >= 512 blocks, 0.57-0.85x
// where it does not).)
template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int GATHER, int TM, int TN, bool PIPE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const TcGemmParams pin, const int splits, const int order,
                                                      const int late_epi) {
  // split-K (blockIdx.y = slice of the K loop): the block emits its raw fp32 partial tile into the
  // workspace -- the plain epilogue with every fused term switched off -- and splitk_reduce_kernel
  // finishes the job
  TcGemmParams p = pin;
  if (splits > 1) {
    p.c = reinterpret_cast<float*>(pin.workspace) + (int64_t)blockIdx.y * pin.m * pin.n;
    p.ldc = pin.n;
    p.out_f32 = 1;
    p.bias = nullptr; p.row_bias = nullptr; p.residual = nullptr;
    p.act = TC_ACT_NONE; p.alpha = 1.f; p.out_scale = 1.f;
  }
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;                       // 32 KiB per stage at 128x128
  constexpr int EPI_BYTES = BM * BN * 4;
  constexpr int SMEM_BYTES = 2 * STAGE_BYTES > EPI_BYTES ? 2 * STAGE_BYTES : EPI_BYTES;
  constexpr int RA = BM / 32, RB = BN / 32;                             // loader rows per thread
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];     // reused by the epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // same value, provably wave-uniform (LDS-DMA base goes in M0)

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  int tile_m, tile_n;
  tc_tile_of_block(blockIdx.x, tiles_m, tiles_n, order, tile_m, tile_n);
  if (tile_m >= tiles_m) return;

  const int64_t bz = blockIdx.z;
  const tc_rsrc_t w_rsrc = make_rsrc(reinterpret_cast<const bf16_t*>(p.w) + bz * p.stride_w, tc_w_extent(p));

  // ---- loader geometry: thread -> (row lrow + 32 i, 16-byte chunk) of both tiles
  // The tile goes global -> LDS directly (buffer_load_dwordx4 ... lds).  One wave instruction fills
  // 1 KiB = 8 tile rows, lane l landing at byte 16 l of the piece, i.e. at (row l>>3, physical chunk
  // l&7): the XOR swizzle is therefore applied to the SOURCE -- the lane fetches the logical chunk whose
  // swizzled home is its own slot.  (row>>1)&7 does not depend on i because rows step by 32.
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  AGather<GATHER, RA> ag;
  ag.init(p, tile_m * BM, lrow, 32, chunk);
  const tc_rsrc_t a_rsrc = tc_a_rsrc(p, bz, ag.row_lo);       // block-relative: 31-bit offsets span one tile's rows
  uint32_t b_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = tile_n * BN + lrow + 32 * i;
    b_voff[i] = n < p.n ? (uint32_t)((int64_t)n * p.ldw * 2 + chunk * 16) : TC_OOB;
  }
  const bool k_ragged = (p.k & (BK - 1)) != 0;      // linear layers only (conv K is a multiple of 64)
#if defined(TC_ABLATE) && (TC_ABLATE & 2)
  const int kb_first = blockIdx.y * (((p.k + BK - 1) / BK + splits - 1) / splits);
#endif

  // issue the loads of K-step kb into `stage`: scalar K offset, per-lane row offsets fixed for the whole
  // block, out-of-range rows / taps / K-tails write zeros -- no staging registers, no ds_write pass, and
  // nothing here costs VALU in the steady state
  auto load_tile = [&](int kb, int stage) {
#if defined(TC_ABLATE) && (TC_ABLATE & 2)      // ... without the tile loads of the steady state
    if (kb >= kb_first + 2) return;
#endif
    const int k0 = kb * BK;
    uint32_t a_voff[RA], a_soff;
    ag.offsets(p, k0, chunk, a_voff, a_soff);
    // OR-ing TC_OOB into an offset keeps it out of range: a K-tail chunk is zero-filled without a branch
    const uint32_t kill = (k_ragged && (k0 + chunk * 8 >= p.k)) ? TC_OOB : 0u;
    char* sa = smem + stage * STAGE_BYTES + wave_u * 1024;
    char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      glds16(w_rsrc, sb + i * 4096, b_voff[i] | kill, (uint32_t)k0 * 2u);
#pragma unroll
    for (int i = 0; i < RA; ++i)
      glds16(a_rsrc, sa + i * 4096, a_voff[i] | kill, a_soff);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  // MFMA fragments are double-buffered in registers: the ds_reads of K-slice kk+1 are issued BEFORE the
  // MFMAs of slice kk, so LDS latency hides under the matrix pipe.
  auto compute = [&](int stage) {
    const char* sa = smem + stage * STAGE_BYTES;
    const char* sb = sa + BM * BK * 2;
    bf16x8 af[2][TM], bf[2][TN];
    auto frags = [&](int kk, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
      const int c = kk * 2 + fhalf;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * 32 * TM + i * 32 + frow, c));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(wn * 32 * TN + j * 32 + frow, c));
    };
    auto mfmas = [&](bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
#if defined(TC_ABLATE) && (TC_ABLATE & 1)      // scripts/ablate_gemm.sh: time the kernel without its MFMAs
      asm volatile("" ::"v"(a[0]), "v"(b[0]));
      return;
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    // sched_barrier(0) pins this order (hipcc would otherwise merge the two fragment sets back into one)
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

  // LDS-DMA data is visible to a ds_read only after the issuing wave's vmcnt wait AND a barrier the
  // reader has passed (MI355X_MICROARCH.md); the same barrier also retires the reads of the stage the
  // next iteration overwrites.
  const int nk_all = (p.k + BK - 1) / BK;
  const int per = (nk_all + splits - 1) / splits;
  const int kb0 = blockIdx.y * per;
  const int nk = min(nk_all, kb0 + per);          // this block runs K-steps [kb0, nk); possibly none
  // epilogue operands from global memory are requested early (gemm_epilogue.h: EpiPrefetch): the bias here, the
  // residual rows in front of the last K-step -- both land under the MFMAs
  const int n_out = p.act == TC_ACT_GEGLU ? p.n / 2 : p.n;
  const bool fast_epi = (n_out & 7) == 0;
  const bool geglu = TN == 2 && p.act == TC_ACT_GEGLU;
  EpiPrefetch pre;
  pre.have_res = false;
  const bool early = fast_epi && !late_epi;        // late_epi (TC_GEMM_EPI_LATE=1, A/B runs): everything inside the epilogue
  if (early) {
    if (geglu) epi_load_bias<true, BM, BN>(p, tid, tile_n, pre);
    else epi_load_bias<false, BM, BN>(p, tid, tile_n, pre);
  }
  if (PIPE) {
    if (kb0 < nk) {
      load_tile(kb0, 0);
      if (kb0 + 1 < nk) load_tile(kb0 + 1, 1);
      for (int kb = kb0; kb < nk; ++kb) {
        const int st = (kb - kb0) & 1;
        // stage st has landed: this wave's pieces (the RA + RB requests of the other stage may stay in flight), then
        // everybody's (barrier)
        if (kb + 1 < nk) gemm_wait_vmcnt<RA + RB>();
        else gemm_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + 1 == nk && early && !geglu && p.residual) epi_load_residual<BM, BN>(p, tid, tile_m, tile_n, bz, pre);
        compute(st);
        if (kb + 2 < nk) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();          // every wave is done reading stage st (its fragments are in registers)
          load_tile(kb + 2, st);
        }
      }
      __syncthreads();                           // the epilogue reuses the stage buffers
    }
  } else if (kb0 < nk) {
    load_tile(kb0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = kb0; kb < nk; ++kb) {
      if (kb + 1 < nk) load_tile(kb + 1, (kb + 1 - kb0) & 1);
      else if (early && !geglu && p.residual) epi_load_residual<BM, BN>(p, tid, tile_m, tile_n, bz, pre);
      compute((kb - kb0) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> LDS fp32 [BM][BN] -> row vectors
  float* cs = reinterpret_cast<float*>(smem);
#if defined(TC_ABLATE) && (TC_ABLATE & 4)      // ... without the epilogue (one guarded store keeps the accumulators live)
  {
    float t = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e30f) reinterpret_cast<float*>(p.c)[tid] = t;
    return;
  }
#endif
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        const int col = wn * 32 * TN + j * 32 + frow;
        cs[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();

  if (fast_epi && !early) {
    if (geglu) epi_load_bias<true, BM, BN>(p, tid, tile_n, pre);
    else epi_load_bias<false, BM, BN>(p, tid, tile_n, pre);
  }
  if (!fast_epi) epilogue_tail<BM, BN>(p, cs, tid, tile_m, tile_n, bz);
  else if (geglu) {
    if (p.alpha == 1.f && p.out_scale == 1.f) epilogue_fast<true, BM, BN, true>(p, cs, tid, tile_m, tile_n, bz, pre);
    else epilogue_fast<true, BM, BN, false>(p, cs, tid, tile_m, tile_n, bz, pre);
  }
  else if (TM == 2 && TN == 2 && pin.gn_part && splits == 1) {
    // GroupNorm statistics of this 128 x 128 tile (ABI 9): per-thread sums over its 8 rows -> lanes of equal column
    // group (lane, lane ^ 16, lane ^ 32) -> the four waves through LDS, in wave order (no atomics: bit-reproducible)
    EpiStats st;
#pragma unroll
    for (int e = 0; e < 8; ++e) { st.s[e] = 0.f; st.q[e] = 0.f; }
    epilogue_fast<false, BM, BN, false, true>(p, cs, tid, tile_m, tile_n, bz, pre, &st);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      st.s[e] += __shfl_xor(st.s[e], 16, 64); st.s[e] += __shfl_xor(st.s[e], 32, 64);
      st.q[e] += __shfl_xor(st.q[e], 16, 64); st.q[e] += __shfl_xor(st.q[e], 32, 64);
    }
    __syncthreads();                                // every thread is done with the fp32 tile
    float* red = reinterpret_cast<float*>(smem);    // [wave][2][BN]
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * BN + lane * 8 + e] = st.s[e];
        red[(wave * 2 + 1) * BN + lane * 8 + e] = st.q[e];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int n = tile_n * BN + tid;
      if (n < p.n) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 2 + 0) * BN + tid]; b += red[(w * 2 + 1) * BN + tid]; }
        float* o = pin.gn_part + (int64_t)tile_m * 2 * p.n + n;
        o[0] = a;
        o[p.n] = b;
      }
    }
  }
  else if (p.alpha == 1.f && p.out_scale == 1.f && p.act == TC_ACT_NONE && !p.row_bias)
    epilogue_fast<false, BM, BN, true>(p, cs, tid, tile_m, tile_n, bz, pre);
  else epilogue_fast<false, BM, BN, false>(p, cs, tid, tile_m, tile_n, bz, pre);
}

// Sum the split-K partial tiles (fixed order: bit-reproducible) and apply the epilogue of `p`:
// one thread per 8 consecutive output columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const TcGemmParams p, const int splits) {
  const int vpr = p.n >> 3;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= (int64_t)p.m * vpr) return;
  const int m = (int)(v / vpr), n0 = (int)(v - (int64_t)m * vpr) * 8;
  const float* ws = reinterpret_cast<const float*>(p.workspace) + (int64_t)m * p.n + n0;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(ws + (int64_t)s * p.m * p.n);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(ws + (int64_t)s * p.m * p.n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[e] += lo[e]; x[4 + e] += hi[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = x[e] * p.alpha;
    if (p.bias) t += p.bias[n0 + e];
    if (p.row_bias) t += p.row_bias[(int64_t)(m / p.row_div) * p.ldrb + n0 + e];
    x[e] = apply_act(t, p.act) * p.out_scale;
  }
  if (p.residual) {
    float rf[8];
    unpack8(*reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.residual) + (int64_t)m * p.ldr + n0), rf);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += rf[e];
  }
  if (p.out_f32) {
    float* op = reinterpret_cast<float*>(p.c) + (int64_t)m * p.ldc + n0;
    *reinterpret_cast<f32x4*>(op) = f32x4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<f32x4*>(op + 4) = f32x4{x[4], x[5], x[6], x[7]};
  } else {
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.c) + (int64_t)m * p.ldc + n0) = pack8(x);
  }
}

}  // namespace

// Split-K factor (1 = no split).  Candidates: one problem (no batching), vector epilogue, no GEGLU, so few
// 128x128 tiles that most of the 512 block slots (2 per CU) would idle, and a K long enough to pay for
// the partial tiles and the reduction pass (~12 us): measured on the M = 1280 layers
// (profiles/r01_v6_gemm_splitk.txt) the 3x3 convolutions (K = 11520 / 23040) gain 1.28x / 1.51x over
// their 64x64-tile launch, while K <= 5120 (linear, temporal conv) loses -- those keep the 64x64 tiles.
// TC_GEMM_SPLITK=n forces n (tuning), 0 disables.
static int tc_gemm_splits(const TcGemmParams& p) {
  const int force = [] { const char* e = getenv("TC_GEMM_SPLITK"); return e ? atoi(e) : -1; }();     // per call (sweeps)
  const int batch = p.batch > 0 ? p.batch : 1;
  if (force == 0 || batch != 1 || p.act == TC_ACT_GEGLU || (p.n & 7) != 0) return 1;
  const int nk = (p.k + BK - 1) / BK;
  const int64_t tiles = (int64_t)((p.n + 127) / 128) * ((p.m + 127) / 128);
  if (tiles >= 200 || (force < 0 && nk < 128)) return 1;
  int s = force > 0 ? force : (int)(512 / tiles);
  if (s > 8) s = 8;
  if (s > nk / 8) s = nk / 8;
  // a power of two: at 100 tiles (the level-3 convolutions) 4 slices measured 1.33-1.41x faster than 5
  // (profiles/r03_gemm_autotune_unet.txt: 61.5 vs 82.3 us at K = 11520, 101.9 vs 143.9 us at K = 23040)
  if (force <= 0) while (s & (s - 1)) s &= s - 1;
  return s < 2 ? 1 : s;
}

extern "C" int64_t tc_gemm_workspace(const TcGemmParams* p) {
  if (!p || p->m <= 0 || p->n <= 0 || p->k <= 0) return 0;
  const int s = tc_gemm_splits(*p);
  return s > 1 ? (int64_t)s * p->m * p->n * (int64_t)sizeof(float) : 0;
}

// Tile family for the 4-wave kernel: (64 tm) x (64 tn).  The sweep (profiles/r01_v6_gemm_tile_sweep.txt)
// has 128x128 ahead on every UNet/decoder shape -- 128x64 and 64x128 lose 5-45 % to their higher LDS/L1
// traffic per FLOP even where they would fill the CUs more evenly -- except the lowest-resolution layers
// (M = 1280 rows), whose 100 128-tiles leave most of the 256 CUs idle: those take 64x64 (+32-37 %).
static void tc_gemm_pick_tile(int m, int n, int batch, bool geglu, int* tm, int* tn) {
  const int64_t big_tiles = (int64_t)((n + 127) / 128) * ((m + 127) / 128) * batch;
  const bool small = !geglu && big_tiles < 384;
  *tm = small ? 1 : 2;
  *tn = small ? 1 : 2;
  // the 3- / 4-channel output convolutions: a 64-column tile wastes half as many MFMAs on padding (measured,
  // profiles/r03_gemm_autotune_*.txt: decoder conv_out 1362 -> 701 us with 128x64, UNet out 106 -> 67 us with 64x64)
  if (!geglu && n <= 64) {
    *tn = 1;
    *tm = m >= 262144 ? 2 : 1;
  }
}

// ABI 9: row-block height of gn_part under the routing tc_gemm_bf16 would take (same order of the same questions)
extern "C" int tc_gemm_gn_rows(const TcGemmParams* pp) {
  if (!pp) return 0;
  const TcGemmParams& p = *pp;
  const int batch = p.batch > 0 ? p.batch : 1;
  if (batch != 1 || p.act == TC_ACT_GEGLU || p.out_f32 || (p.n & 7) != 0 || p.a_norm || p.m <= 0 || p.n <= 0 || p.k <= 0) return 0;
  if (getenv("TC_GEMM_TILE") && getenv("TC_GEMM_TILE")[0]) return 0;          // forced tile families: tuning runs only
  if (const char* e = getenv("TC_GN_PART")) { if (e[0] == '0') return 0; }    // A/B switch: never emit
  if (tc_gemm_ws_try(p, batch, nullptr, true)) return 0;
  // (no question to the halo-patch kernel: it emits no statistics, and tc_gemm_bf16 sets it aside for a call that carries
  // gn_part -- so TC_GN_PART=1 measures what profiles/r04_gn_part_ab.txt measured, producer statistics from every
  // 160 / 128-tile convolution, and not only from the few the halo route leaves over: ADVICE r5)
  if (tc_gemm8_try(p, batch, nullptr, true)) return 0;
  if (tc_gemm_tile16_try(p, batch, nullptr, true)) return 160;
  if (tc_gemm_wide_try(p, batch, nullptr, false, true)) return 0;
  if (p.workspace && tc_gemm_splits(p) > 1 &&
      p.workspace_bytes >= (int64_t)tc_gemm_splits(p) * p.m * p.n * (int64_t)sizeof(float)) return 0;
  int tm, tn;
  tc_gemm_pick_tile(p.m, p.n, batch, false, &tm, &tn);
  return (tm == 2 && tn == 2) ? 128 : 0;
}

extern "C" int tc_gemm_bf16(const TcGemmParams* pp, void* stream) {
  if (!pp) return TC_EINVAL;
  const TcGemmParams& p = *pp;
  if (!p.a || !p.w || !p.c || p.m <= 0 || p.n <= 0 || p.k <= 0) return TC_EINVAL;
  if (!tc_aligned16(p.a) || !tc_aligned16(p.w) || !tc_aligned16(p.c)) return TC_EALIGN;
  if (p.residual && !tc_aligned16(p.residual)) return TC_EALIGN;
  if ((p.k & 7) || (p.lda & 7) || (p.ldw & 7)) return TC_EALIGN;
  if (p.ldw < p.k) return TC_ESHAPE;
  const bool geglu = p.act == TC_ACT_GEGLU;
  const int n_out = geglu ? p.n / 2 : p.n;
  if ((n_out & 7) == 0) {
    if (p.out_f32 ? (p.ldc & 3) : (p.ldc & 7)) return TC_EALIGN;
    if (p.residual && (p.ldr & 7)) return TC_EALIGN;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15u)) return TC_EALIGN;
    if (p.row_bias && ((reinterpret_cast<uintptr_t>(p.row_bias) & 15u) || (p.ldrb & 3))) return TC_EALIGN;
  }
  if (p.ldc < n_out || (p.residual && p.ldr < n_out)) return TC_ESHAPE;
  if (geglu && ((p.n % 128) != 0 || p.row_bias || p.residual)) return TC_ESHAPE;
  if (p.row_bias && (p.row_div <= 0 || p.ldrb < p.n)) return TC_EINVAL;
  if (p.act < TC_ACT_NONE || p.act > TC_ACT_GEGLU) return TC_EINVAL;
  const int batch = p.batch > 0 ? p.batch : 1;
  if (p.gather == TC_GATHER_LINEAR) {
    if (p.lda < p.k) return TC_ESHAPE;
  } else if (p.gather == TC_GATHER_CONV3x3 || p.gather == TC_GATHER_CONVT3) {
    const int taps = p.gather == TC_GATHER_CONV3x3 ? 9 : 3;
    if (p.cin <= 0 || (p.cin % 64) != 0 || p.k != taps * p.cin || p.lda < p.cin) return TC_ESHAPE;
    if (p.frames <= 0 || p.h_out <= 0 || p.w_out <= 0) return TC_ESHAPE;
    if ((int64_t)p.frames * p.h_out * p.w_out != p.m) return TC_ESHAPE;
    if (p.gather == TC_GATHER_CONV3x3) {
      if (p.h_in <= 0 || p.w_in <= 0 || (p.stride != 1 && p.stride != 2)) return TC_ESHAPE;
      if (p.upsample && p.stride != 1) return TC_ESHAPE;
      const int hvv = p.upsample ? 2 * p.h_in : p.h_in, wvv = p.upsample ? 2 * p.w_in : p.w_in;
      if (p.pad != 0 && p.pad != 1) return TC_ESHAPE;
      // pad = 1: symmetric padding 1; pad = 0: one trailing row/column of zeros only (0,1,0,1)
      const int extra = p.pad == 1 ? 2 : 1;
      if ((hvv + extra - 3) / p.stride + 1 != p.h_out || (wvv + extra - 3) / p.stride + 1 != p.w_out) return TC_ESHAPE;
    } else {
      if (p.t_len <= 0 || (p.frames % p.t_len) != 0) return TC_ESHAPE;
    }
  } else {
    return TC_EINVAL;
  }
  if (!tc_gemm_offsets_fit(p)) return TC_ESHAPE;          // buffer-load offsets are 31-bit
  if (p.gn_part && tc_gemm_gn_rows(&p) == 0) return TC_ESHAPE;          // the kernel of this problem emits no statistics
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // TC_GEMM_TILE = wide | 22 | 21 | 12 | 11 forces one tile family (tuning / A-B runs); default: heuristic
  const int force = [] {                                   // per call (scripts/gemm_autotune.py sweeps it in one process)
    const char* e = getenv("TC_GEMM_TILE");
    if (!e) return 0;
    if (e[0] == 'w') return 1;
    if (e[0] == 'b') return 22;
    if (e[0] == 's') return 11;
    const int v = atoi(e);
    return (v == 22 || v == 21 || v == 12 || v == 11) ? v : 0;
  }();
  if (force == 0 && tc_gemm_ws_try(p, batch, s)) {                     // K = 320 linear layers of level 0: W in registers
    TC_LAUNCH_CHECK();
    return TC_OK;
  }
  if (p.a_norm) return TC_ESHAPE;                                      // only the weight-stationary kernel normalises A rows
  if (force == 0 && !p.gn_part) {                                      // tap-reuse patches: the default route of the 3x3 convolutions of levels 0-2 (TC_CONV_HALO=0: never); a call that asks for producer statistics keeps the implicit GEMM, whose epilogue emits them
    const int r = tc_conv_halo_try(p, batch, s);
    if (r < 0) return TC_ESHAPE;                                       // strict mode (tests): a convolution it could not take
    if (r > 0) {
      TC_LAUNCH_CHECK();
      return TC_OK;
    }
  }
  if (force == 0 && tc_gemm8_try(p, batch, s)) {                       // long-K / wide-N problems: 8-wave 256x256 ping-pong
    TC_LAUNCH_CHECK();
    return TC_OK;
  }
  if (force == 0 && tc_gemm_tile16_try(p, batch, s)) {                 // widths 320 k at levels 0 / 1: 160x160 tiles
    TC_LAUNCH_CHECK();
    return TC_OK;
  }
  if (force <= 1 && tc_gemm_wide_try(p, batch, s, force == 1)) {       // big-M layers: 256-row tiles
    TC_LAUNCH_CHECK();
    return TC_OK;
  }
  int tm = 2, tn = 2;
  int splits = (p.workspace && force == 0) ? tc_gemm_splits(p) : 1;
  if (splits > 1 && p.workspace_bytes < (int64_t)splits * p.m * p.n * (int64_t)sizeof(float)) splits = 1;
  if (splits > 1 && !tc_aligned16(p.workspace)) return TC_EALIGN;
  if (force > 1) {
    tm = force / 10;
    tn = geglu ? 2 : force % 10;
  } else if (splits == 1) {
    tc_gemm_pick_tile(p.m, p.n, batch, geglu, &tm, &tn);
  }
  const int bm = 64 * tm, bn = 64 * tn;
  const int tiles_n = (p.n + bn - 1) / bn;
  const int tiles_m = (p.m + bm - 1) / bm;
  const bool nmajor = tc_gemm_nmajor(p);
  const int64_t nblk = nmajor ? 8 * (((int64_t)tiles_m * tiles_n + 7) / 8) : (int64_t)tiles_n * 8 * ((tiles_m + 7) / 8);
  if (nblk > 0x7fffffffLL || batch > 65535) return TC_ESHAPE;
  dim3 grid((unsigned)nblk, (unsigned)splits, (unsigned)batch), block(256);
  const int order = nmajor ? -1 : tc_gemm_tile_order(p, tiles_n);
  // TC_GEMM_PIPE = 0: the one-K-step-in-flight loop; 1 (default): two in flight (read per call: A/B in one process)
  const bool pipe = [] { const char* e = getenv("TC_GEMM_PIPE"); return !(e && e[0] == '0'); }();
  const int late_epi = [] { const char* e = getenv("TC_GEMM_EPI_LATE"); return (e && e[0] == '1') ? 1 : 0; }();
#define TC_LAUNCH_GEMM_P(G, P)                                                                      \
  do {                                                                                              \
    if (tm == 2 && tn == 2) hipLaunchKernelGGL((gemm_kernel<G, 2, 2, P>), grid, block, 0, s, p, splits, order, late_epi);   \
    else if (tm == 2) hipLaunchKernelGGL((gemm_kernel<G, 2, 1, P>), grid, block, 0, s, p, splits, order, late_epi);         \
    else if (tn == 2) hipLaunchKernelGGL((gemm_kernel<G, 1, 2, P>), grid, block, 0, s, p, splits, order, late_epi);         \
    else hipLaunchKernelGGL((gemm_kernel<G, 1, 1, P>), grid, block, 0, s, p, splits, order, late_epi);                      \
  } while (0)
#define TC_LAUNCH_GEMM(G)                                                                           \
  do {                                                                                              \
    if (pipe) TC_LAUNCH_GEMM_P(G, true);                                                            \
    else TC_LAUNCH_GEMM_P(G, false);                                                                \
  } while (0)
  switch (p.gather) {
    case TC_GATHER_LINEAR: TC_LAUNCH_GEMM(TC_GATHER_LINEAR); break;
    case TC_GATHER_CONV3x3: TC_LAUNCH_GEMM(TC_GATHER_CONV3x3); break;
    default: TC_LAUNCH_GEMM(TC_GATHER_CONVT3); break;
  }
#undef TC_LAUNCH_GEMM
#undef TC_LAUNCH_GEMM_P
  TC_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t vecs = (int64_t)p.m * (p.n >> 3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, p, splits);
    TC_LAUNCH_CHECK();
  }
  return TC_OK;
}
