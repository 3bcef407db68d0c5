// Layout converters at the module boundary, embedding helpers, the AE3DConv temporal tail
// and the fused DDIM / classifier-free-guidance step.  All HBM- or latency-bound, gfx950.
#include "common.h"

namespace {

// (B, C, T, HW) fp32 [x0 | x1 on C] -> rows [(b t hw), c_pad] bf16, zero padded.
// One thread per (row, 8-channel vector); reads are strided over C (small C: 3..8) and
// coalesced over hw across the wave.
__global__ void nchw_to_rows_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1, int c1,
                                    bf16_t* __restrict__ out, int c_pad, int nb, int t, int hw, float scale) {
  const int vpr = c_pad >> 3;
  const int64_t total = (int64_t)nb * t * hw * vpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // hw fastest so that a wave reads contiguous floats of one channel plane
    const int p = (int)(i % hw);
    int64_t r = i / hw;
    const int v = (int)(r % vpr);
    r /= vpr;
    const int tt = (int)(r % t);
    const int b = (int)(r / t);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = v * 8 + e;
      float val = 0.f;
      if (ch < c0) val = x0[(((int64_t)b * c0 + ch) * t + tt) * hw + p];
      else if (ch < c0 + c1) val = x1[(((int64_t)b * c1 + (ch - c0)) * t + tt) * hw + p];
      f[e] = val * scale;
    }
    const int64_t row = ((int64_t)b * t + tt) * hw + p;
    *reinterpret_cast<u32x4*>(out + row * c_pad + v * 8) = pack8(f);
  }
}

template <bool SRC_F32>
__global__ void rows_to_nchw_kernel(const void* __restrict__ src, int ld, float* __restrict__ out, int c, int nb,
                                    int t, int hw) {
  const int64_t total = (int64_t)nb * c * t * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    int64_t r = i / hw;
    const int tt = (int)(r % t);
    r /= t;
    const int ch = (int)(r % c);
    const int b = (int)(r / c);
    const int64_t row = ((int64_t)b * t + tt) * hw + p;
    float v;
    if (SRC_F32) v = reinterpret_cast<const float*>(src)[row * ld + ch];
    else v = (float)reinterpret_cast<const bf16_t*>(src)[row * ld + ch];
    out[i] = v;
  }
}

// The same conversion for wide bf16 rows (the first-stage encoder's 128..512-channel hidden states): the
// per-element kernel above reads one 2-byte value per lane at a row stride -- 64 cache lines per wave load.  Here
// a block moves a 64-pixel x 64-channel tile through LDS: 16-byte row-vector reads, 256-byte plane writes.
__global__ __launch_bounds__(256) void rows_to_nchw_tiled_kernel(const bf16_t* __restrict__ src, int ld, float* __restrict__ out,
                                                                 int c, int t, int hw) {
  __shared__ float tile[64][65];                       // [channel][pixel], padded: both phases conflict-free
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int bt = blockIdx.z, b = bt / t, tt = bt - b * t;
  const bf16_t* rows = src + ((int64_t)bt * hw + p0) * ld + c0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = (tid >> 3) + 32 * i, v = tid & 7;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p0 + pix < hw && c0 + v * 8 < c) unpack8(*reinterpret_cast<const u32x4*>(rows + (int64_t)pix * ld + v * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[v * 8 + e][pix] = f[e];
  }
  __syncthreads();
  const int pix = tid & 63;
  if (p0 + pix >= hw) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = (tid >> 6) + 4 * i;
    if (c0 + ch < c) out[(((int64_t)b * c + c0 + ch) * t + tt) * hw + p0 + pix] = tile[ch][pix];
  }
}

__global__ void concat_rows_kernel(const bf16_t* __restrict__ a, int ca, const bf16_t* __restrict__ b, int cb,
                                   bf16_t* __restrict__ out, int64_t rows) {
  const int va = ca >> 3, vb = cb >> 3, vt = va + vb;
  const int64_t total = rows * vt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / vt;
    const int v = (int)(i - r * vt);
    const u32x4 val = v < va ? *reinterpret_cast<const u32x4*>(a + r * ca + v * 8)
                             : *reinterpret_cast<const u32x4*>(b + r * cb + (v - va) * 8);
    *reinterpret_cast<u32x4*>(out + r * (int64_t)(ca + cb) + v * 8) = val;
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int n, int dim, int ld) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld) return;
  const int r = i / ld, col = i - r * ld;
  float v = 0.f;
  if (col < 2 * half) {
    const int j = col < half ? col : col - half;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
    const float arg = t[r] * freq;
    v = col < half ? cosf(arg) : sinf(arg);
  }
  out[i] = (bf16_t)v;
}

__global__ void timestep_embedding_i64_kernel(const int64_t* __restrict__ t, bf16_t* __restrict__ out, int n, int dim, int ld) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld) return;
  const int r = i / ld, col = i - r * ld;
  float v = 0.f;
  if (col < 2 * half) {
    const int j = col < half ? col : col - half;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
    const float arg = (float)t[r] * freq;                                    // == t.to(float32) first, then the fp32 product
    v = col < half ? cosf(arg) : sinf(arg);
  }
  out[i] = (bf16_t)v;
}

// dst[k * rows + r] = src[r] in 16-byte vectors: each source vector is read once and written n times
__global__ __launch_bounds__(256) void repeat_rows_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t vecs, int n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 v = src[i];
    for (int k = 0; k < n; ++k) dst[(int64_t)k * vecs + i] = v;
  }
}

__global__ void silu_f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (bf16_t)silu_f(x[i]);
}

// out[b, co, t, p] = bias[co] + sum_{dt, ci} w[co, ci, dt] * rows[(b, t+dt-1, p), ci]
__global__ void time_mix3_kernel(const float* __restrict__ rows, int ld, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ out, int nb, int t, int hw) {
  float wr[27], br[3];
#pragma unroll
  for (int i = 0; i < 27; ++i) wr[i] = w[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) br[i] = bias[i];
  const int64_t total = (int64_t)nb * t * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const int64_t r = i / hw;
    const int tt = (int)(r % t);
    const int b = (int)(r / t);
    float acc[3] = {br[0], br[1], br[2]};
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const int ts = tt + dt - 1;
      if (ts < 0 || ts >= t) continue;
      const float* src = rows + (((int64_t)b * t + ts) * hw + p) * ld;
      const float s0 = src[0], s1 = src[1], s2 = src[2];
#pragma unroll
      for (int co = 0; co < 3; ++co)
        acc[co] += wr[(co * 3 + 0) * 3 + dt] * s0 + wr[(co * 3 + 1) * 3 + dt] * s1 + wr[(co * 3 + 2) * 3 + dt] * s2;
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) out[(((int64_t)b * 3 + co) * t + tt) * hw + p] = acc[co];
  }
}

// ---------------------------------------------------------------------------------------
// DDIM step.  Pass 1: per-sample partial sums (double) of e_cond and of the CFG combination,
// for the unbiased std of rescale_noise_cfg.  Pass 2: finish the reduction in every block's
// prologue and apply the whole update elementwise.
constexpr int DDIM_PARTS = 64;

__device__ __forceinline__ float cfg_combine(float ec, float eu, float scale) { return eu + scale * (ec - eu); }
// three-way guidance, evaluated left to right like ddim_multiplecond.py:236
__device__ __forceinline__ float cfg_combine3(float ec, float eu, float ei, float scale, float scale_img) {
  return eu + scale_img * (ei - eu) + scale * (ec - ei);
}

__global__ __launch_bounds__(256) void ddim_stats_kernel(TcDdimParams p, double* __restrict__ part) {
  __shared__ double red[4][4];
  const int b = blockIdx.y;
  const float* ec = p.e_cond + (int64_t)b * p.n;
  const float* eu = p.e_uncond + (int64_t)b * p.n;
  const float* ei = p.e_uncond_img ? p.e_uncond_img + (int64_t)b * p.n : nullptr;
  double s[4] = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)DDIM_PARTS * 256) {
    const float c = ec[i];
    const float g = ei ? cfg_combine3(c, eu[i], ei[i], p.cfg_scale, p.cfg_img) : cfg_combine(c, eu[i], p.cfg_scale);
    s[0] += c; s[1] += (double)c * c; s[2] += g; s[3] += (double)g * g;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = s[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    part[((int64_t)b * DDIM_PARTS + blockIdx.x) * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void ddim_apply_kernel(TcDdimParams p, const double* __restrict__ part) {
  __shared__ float factor_s;
  const int b = blockIdx.y;
  const bool cfg = p.e_uncond != nullptr;
  const bool resc = cfg && p.guidance_rescale > 0.f;
  if (threadIdx.x == 0) {
    float factor = 1.f;
    if (resc) {
      double s[4] = {0, 0, 0, 0};
      for (int k = 0; k < DDIM_PARTS; ++k)
        for (int j = 0; j < 4; ++j) s[j] += part[((int64_t)b * DDIM_PARTS + k) * 4 + j];
      const double n = (double)p.n;
      const double var_t = (s[1] - s[0] * s[0] / n) / (n - 1.0);
      const double var_c = (s[3] - s[2] * s[2] / n) / (n - 1.0);
      const float std_t = (float)sqrt(var_t > 0 ? var_t : 0.0);
      const float std_c = (float)sqrt(var_c > 0 ? var_c : 0.0);
      factor = std_t / std_c;
    }
    factor_s = factor;
  }
  __syncthreads();
  const float factor = factor_s;
  const int64_t base = (int64_t)b * p.n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * 256) {
    const float x = p.x[base + i];
    float v = p.e_cond[base + i];
    if (cfg) {
      v = p.e_uncond_img ? cfg_combine3(v, p.e_uncond[base + i], p.e_uncond_img[base + i], p.cfg_scale, p.cfg_img)
                         : cfg_combine(v, p.e_uncond[base + i], p.cfg_scale);
      if (resc) v = p.guidance_rescale * (v * factor) + (1.f - p.guidance_rescale) * v;
    }
    const float e_t = p.sqrt_ac * v + p.sqrt_1m_ac * x;
    float x0 = p.sqrt_ac * x - p.sqrt_1m_ac * v;
    x0 *= p.x0_rescale;
    float xp = p.sqrt_a_prev * x0 + p.dir_coef * e_t;
    if (p.noise) xp += p.sigma * p.noise[base + i];
    p.x_prev[base + i] = xp;
    if (p.pred_x0) p.pred_x0[base + i] = x0;
  }
}

// (b, 3, t, hw) fp32 -> (b, t, hw, 3) uint8: one thread per pixel reads its three channel planes
// (coalesced per plane) and writes 3 consecutive bytes.
__global__ __launch_bounds__(256) void video_to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                          int t, int64_t hw, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t frame = i / hw, pix = i - frame * hw;         // frame = b*t + tt
    const int64_t b = frame / t, tt = frame - b * t;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = x[((b * 3 + c) * t + tt) * hw + pix];
      v = fminf(fmaxf(v, -1.f), 1.f);
      v = (v + 1.0f) / 2.0f;
      out[i * 3 + c] = (uint8_t)(v * 255.f);                    // fp32 -> u8 truncates, like Tensor.to(uint8)
    }
  }
}

inline unsigned grid_for(int64_t n, int threads, unsigned cap = 4096) {
  int64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

}  // namespace

extern "C" int tc_nchw_to_rows(const float* x0, int32_t c0, const float* x1, int32_t c1, tc_bf16* out,
                               int32_t c_pad, int32_t b, int32_t t, int32_t hw, float scale, void* stream) {
  if (!x0 || !out || c0 <= 0 || c1 < 0 || (c1 > 0 && !x1) || b <= 0 || t <= 0 || hw <= 0) return TC_EINVAL;
  if ((c_pad & 7) || c_pad < c0 + c1) return TC_ESHAPE;
  if (!tc_aligned16(out)) return TC_EALIGN;
  const int64_t total = (int64_t)b * t * hw * (c_pad >> 3);
  hipLaunchKernelGGL(nchw_to_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x0, c0, x1, c1, reinterpret_cast<bf16_t*>(out), c_pad, b,
                     t, hw, scale);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_rows_to_nchw(const void* src, int32_t src_f32, int32_t ld, float* out, int32_t c, int32_t b,
                               int32_t t, int32_t hw, void* stream) {
  if (!src || !out || c <= 0 || ld < c || b <= 0 || t <= 0 || hw <= 0) return TC_EINVAL;
  const int64_t total = (int64_t)b * c * t * hw;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (src_f32)
    hipLaunchKernelGGL(rows_to_nchw_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, ld, out, c, b, t, hw);
  else if (c >= 32 && (c & 7) == 0 && (ld & 7) == 0 && tc_aligned16(src) && (int64_t)b * t <= 65535 && (c + 63) / 64 <= 65535)
    hipLaunchKernelGGL(rows_to_nchw_tiled_kernel, dim3((hw + 63) / 64, (c + 63) / 64, b * t), dim3(256), 0, s,
                       reinterpret_cast<const bf16_t*>(src), ld, out, c, t, hw);
  else
    hipLaunchKernelGGL(rows_to_nchw_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, ld, out, c, b, t, hw);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_concat_rows(const tc_bf16* a, int32_t ca, const tc_bf16* b, int32_t cb, tc_bf16* out,
                              int64_t rows, void* stream) {
  if (!a || !b || !out || ca <= 0 || cb <= 0 || rows <= 0) return TC_EINVAL;
  if ((ca & 7) || (cb & 7)) return TC_ESHAPE;
  if (!tc_aligned16(a) || !tc_aligned16(b) || !tc_aligned16(out)) return TC_EALIGN;
  const int64_t total = rows * ((ca + cb) >> 3);
  hipLaunchKernelGGL(concat_rows_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(a), ca,
                     reinterpret_cast<const bf16_t*>(b), cb, reinterpret_cast<bf16_t*>(out), rows);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_timestep_embedding(const float* t, tc_bf16* out, int32_t n, int32_t dim, int32_t ld, void* stream) {
  if (!t || !out || n <= 0 || dim <= 0 || ld < dim) return TC_EINVAL;
  const int total = n * ld;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<bf16_t*>(out), n, dim, ld);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_timestep_embedding_i64(const int64_t* t, tc_bf16* out, int32_t n, int32_t dim, int32_t ld, void* stream) {
  if (!t || !out || n <= 0 || dim <= 0 || ld < dim) return TC_EINVAL;
  const int total = n * ld;
  hipLaunchKernelGGL(timestep_embedding_i64_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<bf16_t*>(out), n, dim, ld);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_repeat_rows(const void* src, void* dst, int64_t rows, int32_t row_bytes, int32_t n, void* stream) {
  if (!src || !dst || rows <= 0 || row_bytes <= 0 || n <= 0) return TC_EINVAL;
  if ((row_bytes & 15) || !tc_aligned16(src) || !tc_aligned16(dst)) return TC_EALIGN;
  const int64_t vecs = rows * (row_bytes >> 4);
  hipLaunchKernelGGL(repeat_rows_kernel, dim3(grid_for(vecs, 256, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(dst), vecs, n);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_silu_f32_to_bf16(const float* x, tc_bf16* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return TC_EINVAL;
  hipLaunchKernelGGL(silu_f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<bf16_t*>(y), n);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_time_mix3(const float* rows, int32_t ld, const float* w, const float* bias, float* out, int32_t b,
                            int32_t t, int32_t hw, void* stream) {
  if (!rows || !w || !bias || !out || ld < 3 || b <= 0 || t <= 0 || hw <= 0) return TC_EINVAL;
  const int64_t total = (int64_t)b * t * hw;
  hipLaunchKernelGGL(time_mix3_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), rows, ld, w, bias, out, b, t, hw);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_video_to_u8(const float* x, uint8_t* out, int32_t b, int32_t t, int32_t hw, void* stream) {
  if (!x || !out || b <= 0 || t <= 0 || hw <= 0) return TC_EINVAL;
  const int64_t total = (int64_t)b * t * hw;
  hipLaunchKernelGGL(video_to_u8_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, out, t, (int64_t)hw, total);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int64_t tc_ddim_workspace(int32_t b) { return b > 0 ? (int64_t)b * DDIM_PARTS * 4 * sizeof(double) : 0; }

extern "C" int tc_ddim_step(const TcDdimParams* pp, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!pp) return TC_EINVAL;
  const TcDdimParams& p = *pp;
  if (!p.x || !p.e_cond || !p.x_prev || p.b <= 0 || p.n <= 1) return TC_EINVAL;
  if (p.e_uncond_img && !p.e_uncond) return TC_EINVAL;
  if (!workspace || workspace_bytes < tc_ddim_workspace(p.b)) return TC_EWORKSPACE;
  if (p.b > 65535) return TC_ESHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  if (p.e_uncond && p.guidance_rescale > 0.f) {
    hipLaunchKernelGGL(ddim_stats_kernel, dim3(DDIM_PARTS, p.b), dim3(256), 0, s, p, part);
    TC_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ddim_apply_kernel, dim3(grid_for(p.n, 256, 256), p.b), dim3(256), 0, s, p, part);
  TC_LAUNCH_CHECK();
  return TC_OK;
}

extern "C" int tc_abi_version(void) { return TC_ABI_VERSION; }
#ifndef TC_SRC_DIGEST
#define TC_SRC_DIGEST "unknown"
#endif
// build.py looks for the "src:<digest>" marker in the binary to decide whether the library is current
extern "C" const char* tc_build_info(void) { return "tooncrafter_hip gfx950 " __DATE__ " " __TIME__ " src:" TC_SRC_DIGEST; }
