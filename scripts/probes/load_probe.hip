// How many operand bytes per clock can a CU pull in?  The GEMM kernels' tile-load pattern (a wave instruction = 64 lanes x
// 16 B = 8 rows x 128 B at a row pitch; 5 "A" pieces from block-private rows + 5 "W" pieces from rows every block shares,
// per K-step, 2 blocks of 4 waves per CU, one barrier per step) with NO compute, through
//   mode 0  buffer_load_dwordx4 ... lds (LDS-DMA) for both operands         -- what gemm16.hip does
//   mode 1  buffer_load_dwordx4 -> VGPR -> ds_write_b128 for both
//   mode 2  A by LDS-DMA, W through registers
//   mode 3  both through registers, no LDS write (xor sink)                 -- the vector-memory path alone
//   mode 4  W only by LDS-DMA      mode 5  W only through registers        mode 6  A only by LDS-DMA    mode 7  A only, registers
// build: hipcc --offload-arch=gfx950 -O3 -I tooncrafter_amd/csrc -I include scripts/probes/load_probe.hip -o scripts/bin/load_probe
#include "gemm_common.h"

#include <stdio.h>
#include <vector>

constexpr int ROWS = 160, R = ROWS / 32;
constexpr int STAGE = 2 * ROWS * 128;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const char* a, int64_t a_bytes, int lda, const char* w, int64_t w_bytes, int ldw,
                                                int ksteps, int iters, uint32_t* sink) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const int tid = threadIdx.x, wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = tid >> 3, chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const tc_rsrc_t a_rsrc = make_rsrc(a + (int64_t)blockIdx.x * ROWS * lda, a_bytes - (int64_t)blockIdx.x * ROWS * lda);
  const tc_rsrc_t w_rsrc = make_rsrc(w, w_bytes);
  uint32_t a_voff[R], w_voff[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    a_voff[i] = (uint32_t)((lrow + 32 * i) * lda + chunk * 16);
    w_voff[i] = (uint32_t)((lrow + 32 * i) * ldw + chunk * 16);
  }
  constexpr bool A_DMA = MODE == 0 || MODE == 2 || MODE == 6, A_REG = MODE == 1 || MODE == 3 || MODE == 7;
  constexpr bool W_DMA = MODE == 0 || MODE == 4, W_REG = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5;
  u32x4 acc = {0u, 0u, 0u, 0u};
  constexpr int NP = (A_DMA || A_REG ? R : 0) + (W_DMA || W_REG ? R : 0);   // vector-memory operations per step and wave
  // two steps in flight: step it + 1 is requested before step it is waited for (counted vmcnt), as a pipelined K loop would
  auto request = [&](int it, u32x4 (&ra)[R], u32x4 (&rw)[R]) {
    const int kb = it % ksteps, stage = it & 1;
    const uint32_t soff = (uint32_t)kb * 128u;
    char* sa = smem + stage * STAGE + wave_u * 1024;
    char* sb = sa + ROWS * 128;
    if (W_DMA) {
#pragma unroll
      for (int i = 0; i < R; ++i) glds16(w_rsrc, sb + i * 4096, w_voff[i], soff);
    }
    if (A_DMA) {
#pragma unroll
      for (int i = 0; i < R; ++i) glds16(a_rsrc, sa + i * 4096, a_voff[i], soff);
    }
    if (W_REG) {
#pragma unroll
      for (int i = 0; i < R; ++i) rw[i] = buf_load16(w_rsrc, w_voff[i], soff);
    }
    if (A_REG) {
#pragma unroll
      for (int i = 0; i < R; ++i) ra[i] = buf_load16(a_rsrc, a_voff[i], soff);
    }
  };
  auto retire = [&](int it, u32x4 (&ra)[R], u32x4 (&rw)[R]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");              // this step has landed; the next one may be in flight
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < R; ++i) acc ^= (A_REG ? ra[i] : u32x4{0u, 0u, 0u, 0u}) ^ (W_REG ? rw[i] : u32x4{0u, 0u, 0u, 0u});
    } else {
      char* la = smem + (it & 1) * STAGE + (tid & 63) * 16 + wave_u * 1024;
      if (W_REG) {
#pragma unroll
        for (int i = 0; i < R; ++i) *reinterpret_cast<u32x4*>(la + ROWS * 128 + i * 4096) = rw[i];
      }
      if (A_REG) {
#pragma unroll
        for (int i = 0; i < R; ++i) *reinterpret_cast<u32x4*>(la + i * 4096) = ra[i];
      }
    }
    __builtin_amdgcn_s_barrier();
  };
  u32x4 ra0[R], rw0[R], ra1[R], rw1[R];
  request(0, ra0, rw0);
  for (int it = 0; it < iters; it += 2) {
    request(it + 1, ra1, rw1);
    retire(it, ra0, rw0);
    request(it + 2, ra0, rw0);
    retire(it + 1, ra1, rw1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 3) acc = *reinterpret_cast<u32x4*>(smem + tid * 16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE>
static void run(const char* tag, const char* a, int64_t ab, int lda, const char* w, int64_t wb, int ldw, int ksteps, uint32_t* sink,
                int pieces) {
  const int iters = 3000, blocks = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, a, ab, lda, w, wb, ldw, ksteps, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)blocks * iters * pieces * 4 * 1024.0;     // pieces per wave x 4 waves x 1 KiB
  const double step_us = best * 1e3 / iters;
  printf("mode %d %-44s %8.3f ms  %6.3f us/step  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", MODE, tag, best, step_us,
         bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  const int lda = 640, ldw = 5760, ksteps = 5;                          // level-0 3x3 convolution: C = 320, K = 2880
  const int64_t ab = (int64_t)512 * ROWS * lda + 4096, wb = (int64_t)ROWS * ldw * 2;
  char *a, *w; uint32_t* sink;
  hipMalloc(&a, ab); hipMalloc(&w, wb); hipMalloc(&sink, 64);
  hipMemset(a, 1, ab); hipMemset(w, 2, wb); hipMemset(sink, 0, 64);
  printf("tile-load probe: 512 blocks x 4 waves, 5 A + 5 W pieces of 1 KiB per wave and step, barrier per step, no compute\n");
  run<0>("A + W by LDS-DMA (gemm16's loads)", a, ab, lda, w, wb, ldw, ksteps, sink, 10);
  run<1>("A + W via registers + ds_write_b128", a, ab, lda, w, wb, ldw, ksteps, sink, 10);
  run<2>("A by LDS-DMA, W via registers + ds_write", a, ab, lda, w, wb, ldw, ksteps, sink, 10);
  run<3>("A + W via registers, no LDS write", a, ab, lda, w, wb, ldw, ksteps, sink, 10);
  run<4>("W only by LDS-DMA", a, ab, lda, w, wb, ldw, ksteps, sink, 5);
  run<5>("W only via registers + ds_write", a, ab, lda, w, wb, ldw, ksteps, sink, 5);
  run<6>("A only by LDS-DMA", a, ab, lda, w, wb, ldw, ksteps, sink, 5);
  run<7>("A only via registers + ds_write", a, ab, lda, w, wb, ldw, ksteps, sink, 5);
  return 0;
}
