// PyTorch-ROCm custom-op layer over the C ABI (include/tooncrafter_hip.h):  TORCH_LIBRARY(tooncrafter, ...)
//
// The reference binds ATen / xformers operators from Python (utils/utils.py:27-42 instantiates lvdm classes whose
// forward methods call torch ops); this registers the MI355X kernels as first-class torch operators instead:
//   torch.ops.tooncrafter.gemm / quant_mxfp8 / gemm_mx / attention / attention_temporal / groupnorm(_pf) / layernorm(_pf) / ddim_step /
//   ff_geglu_fused / temporal_attn_fused / temporal_qkv_attn
// with (i) a CUDA(HIP)-key implementation that validates the tensors, allocates the result from the caching allocator,
// picks up the CURRENT stream and calls the same extern "C" entry point the ctypes binding calls, and (ii) a Meta-key
// implementation (shape / dtype inference only) so the ops can be traced, exported and shape-checked without a GPU.
// Host C++ only: the kernels live in libtooncrafter_hip.so, which this library links.
// Built by tooncrafter_amd/build.py (build_torch_ops) with the host compiler; selected with TC_BINDING=torch
// (tooncrafter_amd/torch_ops.py).  The ctypes binding stays the default: DESIGN.md section 1 measures the two
// launch paths as equivalent at these shapes.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "tooncrafter_hip.h"

namespace {

using at::Tensor;
using c10::optional;

void* cur_stream() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed with code ", rc,
              rc < 0 ? " (TC_E*: -1 invalid, -2 alignment, -3 unsupported shape, -4 workspace)" : " (hipError_t)");
}

const tc_bf16* bf(const Tensor& t) { return reinterpret_cast<const tc_bf16*>(t.data_ptr()); }

void check_rows(const Tensor& t, const char* name, at::ScalarType dt = at::kBFloat16) {
  TORCH_CHECK(t.is_cuda(), name, ": expected a CUDA tensor (the product path is GPU-only)");
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1 && t.scalar_type() == dt, name,
              ": expected a 2-D rows tensor with unit column stride and dtype ", dt);
}

// conv = [] (linear) or [kind (1 = 3x3, 2 = t3), cin, frames, t_len, h_in, w_in, h_out, w_out, stride, upsample, pad]
struct GemmGeom { int64_t m, n, n_out, k; };
GemmGeom gemm_geom(const Tensor& a, const Tensor& w, int64_t act, at::IntArrayRef conv) {
  GemmGeom g;
  g.n = w.size(0);
  g.k = w.size(1);
  g.n_out = act == TC_ACT_GEGLU ? g.n / 2 : g.n;
  g.m = conv.empty() ? a.size(0) : conv[2] * conv[6] * conv[7];
  return g;
}

// fills everything of TcGemmParams except a / w / lda / ldw (operand dtype differs between the bf16 and MXFP8 ops)
void fill_gemm(TcGemmParams& p, const Tensor& a, const GemmGeom& g, Tensor& out, const optional<Tensor>& bias,
               const optional<Tensor>& residual, const optional<Tensor>& row_bias, int64_t row_div, int64_t act, double alpha,
               double out_scale, bool out_f32, at::IntArrayRef conv) {
  TORCH_CHECK(conv.empty() || conv.size() == 11, "gemm: conv must be empty or 11 integers");
  p.c = out.data_ptr();
  p.m = (int32_t)g.m; p.n = (int32_t)g.n; p.k = (int32_t)g.k;
  p.ldc = (int32_t)out.stride(0);
  p.alpha = (float)alpha; p.out_scale = (float)out_scale; p.act = (int32_t)act; p.out_f32 = out_f32 ? 1 : 0;
  p.batch = 1;
  if (bias.has_value()) {
    TORCH_CHECK(bias->is_cuda() && bias->scalar_type() == at::kFloat && bias->is_contiguous() && bias->numel() == g.n,
                "gemm: bias must be a contiguous fp32 CUDA [N]");
    p.bias = bias->data_ptr<float>();
  }
  if (row_bias.has_value()) {
    check_rows(*row_bias, "gemm: row_bias", at::kFloat);
    TORCH_CHECK(row_div > 0 && row_bias->size(1) == g.n && row_bias->size(0) * row_div >= g.m, "gemm: row_bias / row_div do not cover M");
    p.row_bias = row_bias->data_ptr<float>(); p.ldrb = (int32_t)row_bias->stride(0); p.row_div = (int32_t)row_div;
  }
  if (residual.has_value()) {
    check_rows(*residual, "gemm: residual");
    TORCH_CHECK(residual->size(1) == g.n_out && residual->size(0) >= g.m, "gemm: residual shape mismatch");
    p.residual = bf(*residual); p.ldr = (int32_t)residual->stride(0);
  }
  if (conv.empty()) {
    p.gather = TC_GATHER_LINEAR;
    TORCH_CHECK(a.size(1) >= g.k, "gemm: A has fewer columns than the weight's K");
  } else {
    p.gather = conv[0] == 1 ? TC_GATHER_CONV3x3 : TC_GATHER_CONVT3;
    p.cin = (int32_t)conv[1]; p.frames = (int32_t)conv[2]; p.t_len = (int32_t)conv[3];
    p.h_in = (int32_t)conv[4]; p.w_in = (int32_t)conv[5]; p.h_out = (int32_t)conv[6]; p.w_out = (int32_t)conv[7];
    p.stride = (int32_t)conv[8]; p.upsample = (int32_t)conv[9]; p.pad = (int32_t)conv[10];
    TORCH_CHECK(a.size(0) >= (int64_t)p.frames * p.h_in * p.w_in && a.size(1) >= p.cin, "gemm: conv source too small for its geometry");
  }
}

Tensor gemm_cuda(const Tensor& a, const Tensor& w, const optional<Tensor>& bias, const optional<Tensor>& residual,
                 const optional<Tensor>& row_bias, int64_t row_div, int64_t act, double alpha, double out_scale,
                 bool out_f32, at::IntArrayRef conv, double a_norm_eps) {
  check_rows(a, "gemm: a");
  check_rows(w, "gemm: w");
  const GemmGeom g = gemm_geom(a, w, act, conv);
  Tensor out = at::empty({g.m, g.n_out}, a.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
  TcGemmParams p = {};
  p.a = bf(a); p.w = bf(w);
  p.lda = (int32_t)a.stride(0); p.ldw = (int32_t)w.stride(0);
  fill_gemm(p, a, g, out, bias, residual, row_bias, row_div, act, alpha, out_scale, out_f32, conv);
  if (a_norm_eps >= 0.0) {                 // ABI 8: LayerNorm of the A rows as a prologue (affine part folded into w / bias)
    p.a_norm = 1; p.a_norm_eps = (float)a_norm_eps;
    TORCH_CHECK(tc_gemm_ws_eligible(&p) == 1, "gemm: a_norm_eps needs a problem the weight-stationary kernel takes");
  }
  Tensor ws;
  const int64_t nbytes = tc_gemm_workspace(&p);
  if (nbytes > 0) {
    ws = at::empty({nbytes}, a.options().dtype(at::kByte));
    p.workspace = ws.data_ptr(); p.workspace_bytes = nbytes;
  }
  check_rc(tc_gemm_bf16(&p, cur_stream()), "tc_gemm_bf16");
  return out;
}

// MXFP8 pair (ABI 7, BASELINE.json configs[4]): bf16 rows -> (e4m3 bytes, E8M0 scales); GEMM over such operands
std::tuple<Tensor, Tensor> quant_mxfp8_cuda(const Tensor& x, int64_t k) {
  check_rows(x, "quant_mxfp8: x");
  TORCH_CHECK(k > 0 && k % 32 == 0 && x.size(1) >= k, "quant_mxfp8: k must be a multiple of 32 and <= the columns of x");
  const int64_t rows = x.size(0), lds = (k + 127) / 128 * 4;
  Tensor q = at::empty({rows, k}, x.options().dtype(at::kByte));
  Tensor s = at::empty({rows, lds}, x.options().dtype(at::kByte));
  check_rc(tc_quant_mxfp8(bf(x), rows, (int32_t)k, (int32_t)x.stride(0), q.data_ptr<uint8_t>(), (int32_t)q.stride(0),
                          s.data_ptr<uint8_t>(), (int32_t)lds, cur_stream()), "tc_quant_mxfp8");
  return {q, s};
}

std::tuple<Tensor, Tensor> quant_mxfp8_meta(const Tensor& x, int64_t k) {
  return {at::empty({x.size(0), k}, x.options().dtype(at::kByte)),
          at::empty({x.size(0), (k + 127) / 128 * 4}, x.options().dtype(at::kByte))};
}

Tensor gemm_mx_cuda(const Tensor& aq, const Tensor& a_scale, const Tensor& wq, const Tensor& w_scale, const optional<Tensor>& bias,
                    const optional<Tensor>& residual, const optional<Tensor>& row_bias, int64_t row_div, int64_t act,
                    double alpha, double out_scale, bool out_f32, at::IntArrayRef conv) {
  check_rows(aq, "gemm_mx: a", at::kByte); check_rows(wq, "gemm_mx: w", at::kByte);
  check_rows(a_scale, "gemm_mx: a_scale", at::kByte); check_rows(w_scale, "gemm_mx: w_scale", at::kByte);
  TORCH_CHECK(a_scale.size(0) == aq.size(0) && w_scale.size(0) == wq.size(0), "gemm_mx: one scale row per operand row");
  const GemmGeom g = gemm_geom(aq, wq, act, conv);
  Tensor out = at::empty({g.m, g.n_out}, aq.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
  TcGemmMxParams px = {};
  px.g.a = reinterpret_cast<const tc_bf16*>(aq.data_ptr()); px.g.w = reinterpret_cast<const tc_bf16*>(wq.data_ptr());
  px.g.lda = (int32_t)aq.stride(0); px.g.ldw = (int32_t)wq.stride(0);
  fill_gemm(px.g, aq, g, out, bias, residual, row_bias, row_div, act, alpha, out_scale, out_f32, conv);
  px.a_scale = a_scale.data_ptr<uint8_t>(); px.lda_s = (int32_t)a_scale.stride(0);
  px.w_scale = w_scale.data_ptr<uint8_t>(); px.ldw_s = (int32_t)w_scale.stride(0);
  check_rc(tc_gemm_mxfp8(&px, cur_stream()), "tc_gemm_mxfp8");
  return out;
}

Tensor gemm_mx_meta(const Tensor& aq, const Tensor&, const Tensor& wq, const Tensor&, const optional<Tensor>&, const optional<Tensor>&,
                    const optional<Tensor>&, int64_t, int64_t act, double, double, bool out_f32, at::IntArrayRef conv) {
  const GemmGeom g = gemm_geom(aq, wq, act, conv);
  return at::empty({g.m, g.n_out}, aq.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
}

Tensor gemm_meta(const Tensor& a, const Tensor& w, const optional<Tensor>&, const optional<Tensor>&, const optional<Tensor>&,
                 int64_t, int64_t act, double, double, bool out_f32, at::IntArrayRef conv, double) {
  const GemmGeom g = gemm_geom(a, w, act, conv);
  return at::empty({g.m, g.n_out}, a.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
}

Tensor attention_cuda(const Tensor& q, const Tensor& k, const Tensor& v, int64_t batch, int64_t heads, int64_t lq, int64_t lk,
                      int64_t kv_bdiv, double scale, const optional<Tensor>& k2, const optional<Tensor>& v2, int64_t lk2,
                      int64_t kv2_bdiv) {
  check_rows(q, "attention: q"); check_rows(k, "attention: k"); check_rows(v, "attention: v");
  const int64_t hd = heads * 64;
  TORCH_CHECK(q.size(0) == batch * lq && q.size(1) == hd && k.size(1) == hd && v.size(1) == hd, "attention: head layout mismatch");
  const int64_t kvb = (batch + kv_bdiv - 1) / kv_bdiv;
  TORCH_CHECK(k.size(0) == kvb * lk && v.size(0) == kvb * lk, "attention: K/V rows != kv_batches * lk");
  Tensor out = at::empty({batch * lq, hd}, q.options());
  TcAttnParams p = {};
  p.q = bf(q); p.k = bf(k); p.v = bf(v); p.o = reinterpret_cast<tc_bf16*>(out.data_ptr());
  p.batch = (int32_t)batch; p.heads = (int32_t)heads; p.lq = (int32_t)lq; p.lk = (int32_t)lk;
  p.q_ss = (int32_t)q.stride(0); p.k_ss = (int32_t)k.stride(0); p.v_ss = (int32_t)v.stride(0); p.o_ss = (int32_t)out.stride(0);
  p.q_sb = lq * q.stride(0); p.k_sb = lk * k.stride(0); p.v_sb = lk * v.stride(0); p.o_sb = lq * out.stride(0);
  p.kv_bdiv = (int32_t)kv_bdiv; p.scale = (float)scale;
  if (k2.has_value()) {
    TORCH_CHECK(v2.has_value() && lk2 > 0 && kv2_bdiv > 0, "attention: second K/V set incomplete");
    check_rows(*k2, "attention: k2"); check_rows(*v2, "attention: v2");
    const int64_t kvb2 = (batch + kv2_bdiv - 1) / kv2_bdiv;
    TORCH_CHECK(k2->size(0) == kvb2 * lk2 && v2->size(0) == kvb2 * lk2 && k2->size(1) == hd && v2->size(1) == hd,
                "attention: second K/V set must be [(batch / kv2_bdiv) * lk2, heads * 64]");
    p.k2 = bf(*k2); p.v2 = bf(*v2); p.lk2 = (int32_t)lk2; p.kv2_bdiv = (int32_t)kv2_bdiv;
    p.k2_ss = (int32_t)k2->stride(0); p.v2_ss = (int32_t)v2->stride(0);
    p.k2_sb = lk2 * k2->stride(0); p.v2_sb = lk2 * v2->stride(0);
  }
  check_rc(tc_attn_d64(&p, cur_stream()), "tc_attn_d64");
  return out;
}

Tensor attention_meta(const Tensor& q, const Tensor&, const Tensor&, int64_t batch, int64_t heads, int64_t lq, int64_t, int64_t,
                      double, const optional<Tensor>&, const optional<Tensor>&, int64_t, int64_t) {
  return at::empty({batch * lq, heads * 64}, q.options());
}

Tensor attention_temporal_cuda(const Tensor& qkv, int64_t b, int64_t t, int64_t hw, int64_t heads, double scale) {
  check_rows(qkv, "attention_temporal: qkv");
  TORCH_CHECK(qkv.is_contiguous() && qkv.size(0) == b * t * hw && qkv.size(1) == 3 * heads * 64,
              "attention_temporal: qkv must be contiguous [b*t*hw, 3*heads*64]");
  Tensor out = at::empty({b * t * hw, heads * 64}, qkv.options());
  check_rc(tc_attn_temporal(bf(qkv), reinterpret_cast<tc_bf16*>(out.data_ptr()), (int32_t)b, (int32_t)t, (int32_t)hw,
                            (int32_t)heads, (float)scale, cur_stream()), "tc_attn_temporal");
  return out;
}

Tensor attention_temporal_meta(const Tensor& qkv, int64_t b, int64_t t, int64_t hw, int64_t heads, double) {
  return at::empty({b * t * hw, heads * 64}, qkv.options());
}

void check_affine(const Tensor& g, const Tensor& b, int64_t c, const char* what) {
  TORCH_CHECK(g.is_cuda() && b.is_cuda() && g.scalar_type() == at::kFloat && b.scalar_type() == at::kFloat &&
              g.is_contiguous() && b.is_contiguous() && g.numel() == c && b.numel() == c, what, ": gamma / beta must be contiguous fp32 CUDA [C]");
}

Tensor groupnorm_cuda(const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t samples, int64_t rows, double eps, bool silu) {
  check_rows(x, "groupnorm: x");
  const int64_t c = x.size(1);
  TORCH_CHECK(x.is_contiguous() && x.size(0) == samples * rows, "groupnorm: x must be contiguous [samples*rows, C]");
  check_affine(gamma, beta, c, "groupnorm");
  Tensor y = at::empty_like(x);
  const int64_t nbytes = tc_groupnorm_workspace((int32_t)samples, (int32_t)rows, (int32_t)c);
  Tensor ws = at::empty({nbytes > 16 ? nbytes : 16}, x.options().dtype(at::kByte));
  check_rc(tc_groupnorm(bf(x), reinterpret_cast<tc_bf16*>(y.data_ptr()), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                        (int32_t)samples, (int32_t)rows, (int32_t)c, (float)eps, silu ? 1 : 0, ws.data_ptr(), nbytes, cur_stream()),
           "tc_groupnorm");
  return y;
}

// ABI 12: the consumer's weights (read-only CUDA tensors, contiguous) as a prefetch list for the norm's extra blocks
TcPrefetch prefetch_of(at::TensorList ts, const char* what) {
  TORCH_CHECK((int64_t)ts.size() <= TC_PREFETCH_MAX, what, ": at most ", TC_PREFETCH_MAX, " prefetch tensors");
  TcPrefetch pf = {};
  for (const Tensor& t : ts) {
    TORCH_CHECK(t.is_cuda() && t.is_contiguous(), what, ": prefetch tensors must be contiguous CUDA tensors");
    pf.ptr[pf.n] = t.data_ptr();
    pf.bytes[pf.n] = (int64_t)t.numel() * (int64_t)t.element_size();
    ++pf.n;
  }
  return pf;
}

Tensor groupnorm_pf_cuda(const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t samples, int64_t rows, double eps, bool silu,
                         at::TensorList prefetch) {
  check_rows(x, "groupnorm_pf: x");
  const int64_t c = x.size(1);
  TORCH_CHECK(x.is_contiguous() && x.size(0) == samples * rows, "groupnorm_pf: x must be contiguous [samples*rows, C]");
  check_affine(gamma, beta, c, "groupnorm_pf");
  const TcPrefetch pf = prefetch_of(prefetch, "groupnorm_pf");
  Tensor y = at::empty_like(x);
  const int64_t nbytes = tc_groupnorm_workspace((int32_t)samples, (int32_t)rows, (int32_t)c);
  Tensor ws = at::empty({nbytes > 16 ? nbytes : 16}, x.options().dtype(at::kByte));
  check_rc(tc_groupnorm_pf(bf(x), reinterpret_cast<tc_bf16*>(y.data_ptr()), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                           (int32_t)samples, (int32_t)rows, (int32_t)c, (float)eps, silu ? 1 : 0, ws.data_ptr(), nbytes, &pf, cur_stream()),
           "tc_groupnorm_pf");
  return y;
}

Tensor layernorm_pf_cuda(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps, at::TensorList prefetch) {
  check_rows(x, "layernorm_pf: x");
  TORCH_CHECK(x.is_contiguous(), "layernorm_pf: x must be contiguous");
  check_affine(gamma, beta, x.size(1), "layernorm_pf");
  const TcPrefetch pf = prefetch_of(prefetch, "layernorm_pf");
  Tensor y = at::empty_like(x);
  check_rc(tc_layernorm_pf(bf(x), reinterpret_cast<tc_bf16*>(y.data_ptr()), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                           (int32_t)x.size(0), (int32_t)x.size(1), (float)eps, &pf, cur_stream()), "tc_layernorm_pf");
  return y;
}

Tensor groupnorm_pf_meta(const Tensor& x, const Tensor&, const Tensor&, int64_t, int64_t, double, bool, at::TensorList) { return at::empty_like(x); }
Tensor layernorm_pf_meta(const Tensor& x, const Tensor&, const Tensor&, double, at::TensorList) { return at::empty_like(x); }

// ---- the level-0 one-launch operators (ABI 9): LayerNorm + GEGLU feed-forward + residual; LayerNorm + temporal self-attention +
// residual (lvdm/modules/attention.py:81-144,225-246,415-442).  ln_eps < 0: x is taken as already normalised.
void check_w(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == dt && t.is_contiguous(), name, ": expected a contiguous CUDA tensor of dtype ", dt);
}

Tensor ff_geglu_fused_cuda(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double ln_eps) {
  check_rows(x, "ff_geglu_fused: x");
  check_w(w1, at::kBFloat16, "ff_geglu_fused: w1"); check_w(w2, at::kBFloat16, "ff_geglu_fused: w2");
  check_w(b1, at::kFloat, "ff_geglu_fused: b1"); check_w(b2, at::kFloat, "ff_geglu_fused: b2");
  const int64_t m = x.size(0), c = x.size(1);
  TORCH_CHECK(w2.dim() == 2 && w1.dim() == 2, "ff_geglu_fused: w1 [2 hidden, c], w2 [c, hidden]");
  const int64_t hidden = w2.size(1);
  TORCH_CHECK(w1.size(0) == 2 * hidden && w1.size(1) == c && w2.size(0) == c && b1.numel() == 2 * hidden && b2.numel() == c,
              "ff_geglu_fused: x [m, c] rows, w1 [2 hidden, c], b1 [2 hidden], w2 [c, hidden], b2 [c]");
  Tensor out = at::empty({m, c}, x.options());
  TcFfParams p{};
  p.x = bf(x); p.w1 = bf(w1); p.b1 = b1.data_ptr<float>(); p.w2 = bf(w2); p.b2 = b2.data_ptr<float>();
  p.out = reinterpret_cast<tc_bf16*>(out.data_ptr());
  p.m = (int32_t)m; p.c = (int32_t)c; p.hidden = (int32_t)hidden; p.ldx = (int32_t)x.stride(0); p.ldo = (int32_t)c;
  p.ln = ln_eps >= 0.0 ? 1 : 0; p.ln_eps = ln_eps >= 0.0 ? (float)ln_eps : 0.f;
  check_rc(tc_ff_geglu_fused(&p, cur_stream()), "tc_ff_geglu_fused");
  return out;
}

Tensor ff_geglu_fused_meta(const Tensor& x, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double) {
  return at::empty({x.size(0), x.size(1)}, x.options());
}

Tensor temporal_attn_fused_cuda(const Tensor& x, const Tensor& wqkv, const Tensor& bqkv, const Tensor& wo, const Tensor& bo, int64_t b,
                                int64_t t, int64_t hw, int64_t heads, double ln_eps, double scale) {
  check_rows(x, "temporal_attn_fused: x");
  check_w(wqkv, at::kBFloat16, "temporal_attn_fused: wqkv"); check_w(wo, at::kBFloat16, "temporal_attn_fused: wo");
  check_w(bqkv, at::kFloat, "temporal_attn_fused: bqkv"); check_w(bo, at::kFloat, "temporal_attn_fused: bo");
  const int64_t m = x.size(0), c = x.size(1);
  TORCH_CHECK(m == b * t * hw && wqkv.dim() == 2 && wqkv.size(0) == 3 * c && wqkv.size(1) == c && wo.dim() == 2 && wo.size(0) == c &&
              wo.size(1) == c && bqkv.numel() == 3 * c && bo.numel() == c && c == heads * 64,
              "temporal_attn_fused: x [b*t*hw, c] rows, wqkv [3c, c], bqkv [3c], wo [c, c], bo [c], c = heads * 64");
  Tensor out = at::empty({m, c}, x.options());
  TcTbParams p{};
  p.x = bf(x); p.wqkv = bf(wqkv); p.bqkv = bqkv.data_ptr<float>(); p.wo = bf(wo); p.bo = bo.data_ptr<float>();
  p.out = reinterpret_cast<tc_bf16*>(out.data_ptr());
  p.b = (int32_t)b; p.t = (int32_t)t; p.hw = (int32_t)hw; p.c = (int32_t)c; p.heads = (int32_t)heads;
  p.ldx = (int32_t)x.stride(0); p.ldo = (int32_t)c;
  p.ln = ln_eps >= 0.0 ? 1 : 0; p.ln_eps = ln_eps >= 0.0 ? (float)ln_eps : 0.f; p.scale = (float)scale;
  check_rc(tc_temporal_attn_fused(&p, cur_stream()), "tc_temporal_attn_fused");
  return out;
}

Tensor temporal_attn_fused_meta(const Tensor& x, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t,
                                int64_t, double, double) {
  return at::empty({x.size(0), x.size(1)}, x.options());
}

// ABI 13: the temporal q / k / v projection and its attention as one launch (csrc/qkv_attn.hip)
Tensor temporal_qkv_attn_cuda(const Tensor& x, const Tensor& wqkv, const optional<Tensor>& bqkv, int64_t b, int64_t t, int64_t hw,
                              int64_t heads, double scale) {
  check_rows(x, "temporal_qkv_attn: x");
  check_w(wqkv, at::kBFloat16, "temporal_qkv_attn: wqkv");
  if (bqkv.has_value()) check_w(*bqkv, at::kFloat, "temporal_qkv_attn: bqkv");
  const int64_t m = x.size(0), c = x.size(1);
  TORCH_CHECK(m == b * t * hw && wqkv.dim() == 2 && wqkv.size(0) == 3 * c && wqkv.size(1) == c && c == heads * 64 &&
              (!bqkv.has_value() || bqkv->numel() == 3 * c),
              "temporal_qkv_attn: x [b*t*hw, c] rows, wqkv [3c, c], bqkv [3c] or None, c = heads * 64");
  Tensor out = at::empty({m, c}, x.options());
  TcTqaParams p{};
  p.x = bf(x); p.wqkv = bf(wqkv); p.bqkv = bqkv.has_value() ? bqkv->data_ptr<float>() : nullptr;
  p.out = reinterpret_cast<tc_bf16*>(out.data_ptr());
  p.b = (int32_t)b; p.t = (int32_t)t; p.hw = (int32_t)hw; p.c = (int32_t)c; p.heads = (int32_t)heads;
  p.ldx = (int32_t)x.stride(0); p.ldo = (int32_t)c;
  p.scale = (float)scale;
  check_rc(tc_temporal_qkv_attn(&p, cur_stream()), "tc_temporal_qkv_attn");
  return out;
}

Tensor temporal_qkv_attn_meta(const Tensor& x, const Tensor&, const optional<Tensor>&, int64_t, int64_t, int64_t, int64_t, double) {
  return at::empty({x.size(0), x.size(1)}, x.options());
}

Tensor layernorm_cuda(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
  check_rows(x, "layernorm: x");
  TORCH_CHECK(x.is_contiguous(), "layernorm: x must be contiguous");
  check_affine(gamma, beta, x.size(1), "layernorm");
  Tensor y = at::empty_like(x);
  check_rc(tc_layernorm(bf(x), reinterpret_cast<tc_bf16*>(y.data_ptr()), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                        (int32_t)x.size(0), (int32_t)x.size(1), (float)eps, cur_stream()), "tc_layernorm");
  return y;
}

Tensor like_meta3(const Tensor& x, const Tensor&, const Tensor&, int64_t, int64_t, double, bool) { return at::empty_like(x); }
Tensor like_meta_ln(const Tensor& x, const Tensor&, const Tensor&, double) { return at::empty_like(x); }

std::tuple<Tensor, Tensor> ddim_step_cuda(const Tensor& x, const Tensor& e_cond, const optional<Tensor>& e_uncond,
                                          const optional<Tensor>& noise, const optional<Tensor>& e_uncond_img, double cfg_scale,
                                          double cfg_img, double guidance_rescale, double sqrt_ac, double sqrt_1m_ac,
                                          double sqrt_a_prev, double dir_coef, double sigma, double x0_rescale) {
  auto ok = [](const Tensor& t) { return t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(); };
  TORCH_CHECK(ok(x) && ok(e_cond) && (!e_uncond.has_value() || ok(*e_uncond)) && (!noise.has_value() || ok(*noise)) &&
              (!e_uncond_img.has_value() || ok(*e_uncond_img)), "ddim_step: contiguous fp32 CUDA tensors");
  Tensor x_prev = at::empty_like(x), x0 = at::empty_like(x);
  TcDdimParams p = {};
  p.x = x.data_ptr<float>(); p.e_cond = e_cond.data_ptr<float>();
  p.e_uncond = e_uncond.has_value() ? e_uncond->data_ptr<float>() : nullptr;
  p.noise = noise.has_value() ? noise->data_ptr<float>() : nullptr;
  p.e_uncond_img = e_uncond_img.has_value() ? e_uncond_img->data_ptr<float>() : nullptr;
  p.x_prev = x_prev.data_ptr<float>(); p.pred_x0 = x0.data_ptr<float>();
  p.b = (int32_t)x.size(0); p.n = x.numel() / x.size(0);
  p.cfg_scale = (float)cfg_scale; p.cfg_img = (float)cfg_img; p.guidance_rescale = (float)guidance_rescale;
  p.sqrt_ac = (float)sqrt_ac; p.sqrt_1m_ac = (float)sqrt_1m_ac; p.sqrt_a_prev = (float)sqrt_a_prev;
  p.dir_coef = (float)dir_coef; p.sigma = (float)sigma; p.x0_rescale = (float)x0_rescale;
  const int64_t nbytes = tc_ddim_workspace(p.b);
  Tensor ws = at::empty({nbytes > 16 ? nbytes : 16}, x.options().dtype(at::kByte));
  check_rc(tc_ddim_step(&p, ws.data_ptr(), nbytes, cur_stream()), "tc_ddim_step");
  return std::make_tuple(x_prev, x0);
}

std::tuple<Tensor, Tensor> ddim_step_meta(const Tensor& x, const Tensor&, const optional<Tensor>&, const optional<Tensor>&,
                                          const optional<Tensor>&, double, double, double, double, double, double, double, double, double) {
  return std::make_tuple(at::empty_like(x), at::empty_like(x));
}

}  // namespace

TORCH_LIBRARY(tooncrafter, m) {
  m.def("gemm(Tensor a, Tensor w, Tensor? bias, Tensor? residual, Tensor? row_bias, int row_div, int act, float alpha, "
        "float out_scale, bool out_f32, int[] conv, float a_norm_eps=-1.0) -> Tensor");
  m.def("quant_mxfp8(Tensor x, int k) -> (Tensor, Tensor)");
  m.def("gemm_mx(Tensor aq, Tensor a_scale, Tensor wq, Tensor w_scale, Tensor? bias, Tensor? residual, Tensor? row_bias, "
        "int row_div, int act, float alpha, float out_scale, bool out_f32, int[] conv) -> Tensor");
  m.def("attention(Tensor q, Tensor k, Tensor v, int batch, int heads, int lq, int lk, int kv_bdiv, float scale, "
        "Tensor? k2, Tensor? v2, int lk2, int kv2_bdiv) -> Tensor");
  m.def("attention_temporal(Tensor qkv, int b, int t, int hw, int heads, float scale) -> Tensor");
  m.def("groupnorm(Tensor x, Tensor gamma, Tensor beta, int samples, int rows, float eps, bool silu) -> Tensor");
  m.def("ff_geglu_fused(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2, float ln_eps) -> Tensor");
  m.def("temporal_attn_fused(Tensor x, Tensor wqkv, Tensor bqkv, Tensor wo, Tensor bo, int b, int t, int hw, int heads, float ln_eps, float scale) -> Tensor");
  m.def("temporal_qkv_attn(Tensor x, Tensor wqkv, Tensor? bqkv, int b, int t, int hw, int heads, float scale) -> Tensor");
  m.def("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps) -> Tensor");
  m.def("groupnorm_pf(Tensor x, Tensor gamma, Tensor beta, int samples, int rows, float eps, bool silu, Tensor[] prefetch) -> Tensor");
  m.def("layernorm_pf(Tensor x, Tensor gamma, Tensor beta, float eps, Tensor[] prefetch) -> Tensor");
  m.def("ddim_step(Tensor x, Tensor e_cond, Tensor? e_uncond, Tensor? noise, Tensor? e_uncond_img, float cfg_scale, "
        "float cfg_img, float guidance_rescale, float sqrt_ac, float sqrt_1m_ac, float sqrt_a_prev, float dir_coef, "
        "float sigma, float x0_rescale) -> (Tensor, Tensor)");
  m.def("abi_version() -> int");
}

TORCH_LIBRARY_IMPL(tooncrafter, CUDA, m) {
  m.impl("gemm", gemm_cuda);
  m.impl("quant_mxfp8", quant_mxfp8_cuda);
  m.impl("gemm_mx", gemm_mx_cuda);
  m.impl("attention", attention_cuda);
  m.impl("attention_temporal", attention_temporal_cuda);
  m.impl("groupnorm", groupnorm_cuda);
  m.impl("layernorm", layernorm_cuda);
  m.impl("groupnorm_pf", groupnorm_pf_cuda);
  m.impl("layernorm_pf", layernorm_pf_cuda);
  m.impl("ddim_step", ddim_step_cuda);
  m.impl("ff_geglu_fused", ff_geglu_fused_cuda);
  m.impl("temporal_attn_fused", temporal_attn_fused_cuda);
  m.impl("temporal_qkv_attn", temporal_qkv_attn_cuda);
}

TORCH_LIBRARY_IMPL(tooncrafter, Meta, m) {
  m.impl("gemm", gemm_meta);
  m.impl("quant_mxfp8", quant_mxfp8_meta);
  m.impl("gemm_mx", gemm_mx_meta);
  m.impl("attention", attention_meta);
  m.impl("attention_temporal", attention_temporal_meta);
  m.impl("groupnorm", like_meta3);
  m.impl("layernorm", like_meta_ln);
  m.impl("groupnorm_pf", groupnorm_pf_meta);
  m.impl("layernorm_pf", layernorm_pf_meta);
  m.impl("ddim_step", ddim_step_meta);
  m.impl("ff_geglu_fused", ff_geglu_fused_meta);
  m.impl("temporal_attn_fused", temporal_attn_fused_meta);
  m.impl("temporal_qkv_attn", temporal_qkv_attn_meta);
}

TORCH_LIBRARY_IMPL(tooncrafter, CompositeExplicitAutograd, m) {
  m.impl("abi_version", []() -> int64_t { return tc_abi_version(); });
}
