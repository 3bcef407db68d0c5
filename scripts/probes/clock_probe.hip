// In-stream clock probe: one thread records the shader-clock counter (s_memtime, clock64()) and the constant 100 MHz
// counter (s_memrealtime, wall_clock64()).  Two probes bracket a stretch of work on the SAME stream:
//   average engine clock over the stretch = d(clock64) / d(wall_clock64) x 100 MHz.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC scripts/probes/clock_probe.hip -o scripts/bin/libclock_probe.so
#include <hip/hip_runtime.h>

extern "C" __global__ void clk_kernel(unsigned long long* out) {
  out[0] = (unsigned long long)clock64();
  out[1] = (unsigned long long)wall_clock64();
}

extern "C" int clk_probe(unsigned long long* out, void* stream) {
  hipLaunchKernelGGL(clk_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), out);
  return (int)hipGetLastError();
}
