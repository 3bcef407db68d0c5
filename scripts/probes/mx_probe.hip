// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales): which K index does byte b of
// lane (l31, half) carry, and whose scale is a lane's scale byte?  Small exactly-representable values, host reference.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* out, int hyp, int use_scale) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  uint8_t ab[32], bb[32];
  for (int b = 0; b < 32; ++b) {
    int kk = hyp == 0 ? half * 32 + b : hyp == 1 ? (b / 16) * 32 + half * 16 + (b % 16) : (b / 8) * 16 + half * 8 + (b % 8);
    ab[b] = A[l31 * 64 + kk];     // A[m][k], lane row m = l31
    bb[b] = B[l31 * 64 + kk];     // B stored [n][k], lane col n = l31
  }
  v8i a, bv;
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    bv[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  // scale VGPR: byte 0 = this lane's E8M0 scale for (row l31, K-block half) under the natural hypothesis
  int sca = use_scale ? sa[l31 * 2 + half] : 127, scb = use_scale ? sb[l31 * 2 + half] : 127;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bv, c, 0, 0, 0, sca, 0, scb);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = c[r];   // [m][n]
}
static float e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float f = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s ? -f : f; }
int main() {
  const uint8_t vals[8] = {0x00, 0x38, 0x40, 0xB8, 0x30, 0x3C, 0xC0, 0x44};   // 0 1 2 -1 0.5 1.5 -2 3
  uint8_t hA[32 * 64], hB[32 * 64], hsa[64], hsb[64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { hA[i] = vals[rand() % 8]; hB[i] = vals[rand() % 8]; }
  for (int i = 0; i < 64; ++i) { hsa[i] = 124 + rand() % 7; hsb[i] = 124 + rand() % 7; }
  uint8_t *dA, *dB, *dsa, *dsb; float* dout;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dout, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
  for (int use_scale = 0; use_scale < 2; ++use_scale) {
    float ref[32 * 32];
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int kk = 0; kk < 64; ++kk) {
        double x = e4m3(hA[m * 64 + kk]), y = e4m3(hB[n * 64 + kk]);
        if (use_scale) { x *= ldexp(1.0, hsa[m * 2 + kk / 32] - 127); y *= ldexp(1.0, hsb[n * 2 + kk / 32] - 127); }
        s += x * y;
      }
      ref[m * 32 + n] = (float)s;
    }
    for (int hyp = 0; hyp < 3; ++hyp) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dout, hyp, use_scale);
      float h[32 * 32]; hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
      double err = 0, nrm = 0; for (int i = 0; i < 1024; ++i) { err += (h[i] - ref[i]) * (h[i] - ref[i]); nrm += ref[i] * ref[i]; }
      printf("scales %s, K-layout hypothesis %d: rel err %.3e  (out[0][0..3] = %g %g %g %g, ref %g %g %g %g)\n", use_scale ? "random" : "1.0",
             hyp, sqrt(err / nrm), h[0], h[1], h[2], h[3], ref[0], ref[1], ref[2], ref[3]);
    }
  }
  return 0;
}
