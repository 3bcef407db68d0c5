#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane-linear: lane L points at elements [4L, 4L+3] (stride_bytes = 8) or at its own row (stride_bytes = row pitch)
  const char* p = reinterpret_cast<const char*>(lds) + lane * stride_bytes;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {8, 32, 128}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d bytes (lane L address = L*%d; values are element indices = byte/2)\n", stride, stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
