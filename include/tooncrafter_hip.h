/*
 * tooncrafter_hip.h -- C ABI of the MI355X (gfx950) kernels behind the ToonCrafter
 * denoising hot path (DDIM sampler -> spatio-temporal UNet -> dual-reference
 * VideoDecoder).
 *
 * The reference (Doubiiu/ToonCrafter) is pure PyTorch: it has no FFI of its own.
 * What it binds instead are third-party kernels (ATen conv/GEMM/norm/softmax,
 * xformers.memory_efficient_attention).  Every entry point below replaces one of
 * those call sites; the reference file:line is given with each.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory, sizes as int/int64, no torch types;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - never allocates, never synchronises, hipGraph-capture safe;
 *   - returns 0 on success, a negative TC_E* code on a bad argument (nothing launched),
 *     or the positive hipError_t of a failed launch;
 *   - activations are bf16 "channels-last rows": a tensor (B,T,H,W,C) is a row-major
 *     matrix [B*T*H*W, C]; weights are bf16 [N, K] row-major (K contiguous).
 */
#ifndef TOONCRAFTER_HIP_H
#define TOONCRAFTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_ABI_VERSION 13

enum {
  TC_OK = 0,
  TC_EINVAL = -1,      /* null pointer / non-positive size */
  TC_EALIGN = -2,      /* pointer or leading dimension not 16-byte aligned */
  TC_ESHAPE = -3,      /* shape outside what the kernel supports */
  TC_EWORKSPACE = -4   /* workspace too small */
};

typedef uint16_t tc_bf16;   /* raw bfloat16 bits */

/* ---- activation applied in the GEMM epilogue ---- */
enum { TC_ACT_NONE = 0, TC_ACT_SILU = 1, TC_ACT_GELU = 2, TC_ACT_GEGLU = 3 };

/* ---- how GEMM rows of A are gathered (implicit-GEMM convolution) ---- */
enum {
  TC_GATHER_LINEAR = 0,   /* A[m, k] = a[m*lda + k]                                  (nn.Linear, 1x1 conv) */
  TC_GATHER_CONV3x3 = 1,  /* 3x3, pad 1, stride `stride`, optional nearest-x2 source  (nn.Conv2d)           */
  TC_GATHER_CONVT3 = 2    /* (3,1,1) over T, pad (1,0,0)                              (nn.Conv3d)           */
};

typedef struct TcGemmParams {
  /* operands */
  const tc_bf16* a;        /* activations, channels-last rows */
  const tc_bf16* w;        /* [N, K] bf16; K = taps*cin, tap-major then channel; for TC_ACT_GEGLU
                              every 32 rows are packed as 16 value rows then their 16 gate rows */
  void* c;                 /* [M, ldc] bf16 (or fp32 when out_f32) */
  const float* bias;       /* [N] fp32 or NULL (same packing as w rows) */
  const float* row_bias;   /* [ceil(M/row_div), N] fp32 or NULL: per-frame embedding add */
  const tc_bf16* residual; /* [M, ldr] bf16 or NULL, added last */
  /* sizes */
  int32_t m, n, k;         /* n, k as stored in w; with GEGLU the output has n/2 columns */
  int32_t lda, ldw, ldc, ldr; /* leading dimensions in elements (multiples of 8); ldw >= k */
  int32_t ldrb;            /* leading dimension of row_bias (>= n) */
  int32_t row_div;         /* rows sharing one row_bias row (H*W) */
  /* epilogue:  v = act(alpha*acc + bias + row_bias) * out_scale + residual */
  float alpha, out_scale;
  int32_t act;
  int32_t out_f32;
  /* gather geometry */
  int32_t gather;
  int32_t cin;             /* channels per tap (multiple of 64 when taps > 1) */
  int32_t frames;          /* B*T images */
  int32_t t_len;           /* T, frames per clip (CONVT3 bounds) */
  int32_t h_out, w_out;    /* output image size: m = (frame*h_out + y)*w_out + x */
  int32_t h_in, w_in;      /* source image size */
  int32_t stride;          /* 1 or 2 */
  int32_t upsample;        /* 1: source is nearest-upsampled x2 on the fly (Upsample + conv fused) */
  int32_t pad;             /* leading (top/left) zero padding of the 3x3 gather: 1 = symmetric pad 1;
                              0 = the VAE encoder's asymmetric (0,1,0,1) pad before its stride-2 conv */
  /* batching (blockIdx.z): element strides, 0 = shared */
  int32_t batch;
  int64_t stride_a, stride_w, stride_c;
  /* ABI 4 -- optional scratch for split-K: low-resolution layers (few output tiles, long K) are cut along K
   * over several blocks per tile, fp32 partial tiles go here and a second kernel sums them in a fixed order
   * and applies the epilogue (bit-reproducible).  NULL / too small = never split. */
  void* workspace;
  int64_t workspace_bytes;
  /* ABI 8 -- LayerNorm of the A rows as a prologue of the product (lvdm/modules/attention.py:225-227 in front of the
   * qkv / GEGLU projections, attention.py:242-246): a_norm = 1 normalises every row over its k columns,
   * (x - mean) * rsqrt(var + a_norm_eps) with fp32 statistics, before it is multiplied; the affine part is the
   * CALLER's job (w[n, :] *= gamma, bias += w @ beta: exact in real arithmetic).  Only problems tc_gemm_ws_eligible
   * accepts can carry it (whole rows of A must sit in one block); tc_gemm_bf16 returns TC_ESHAPE otherwise. */
  int32_t a_norm;
  float a_norm_eps;
  /* ABI 9 -- GroupNorm statistics from the producer (lvdm/basics.py:76-87 normalises what openaimodel3d.py:154,179,
   * 255-266 just wrote): not NULL = besides C the launch emits, for every block of tc_gemm_gn_rows() consecutive output
   * rows and every output column, the sum and the sum of squares of the bf16-ROUNDED outputs, fp32
   * [ceil(m / gn_rows)][2][n], summed in a fixed order (no atomics).  tc_groupnorm_part turns them into (mean, rstd) --
   * the consumer's own statistics pass over the tensor (one of GroupNorm's two reads) disappears.  Only for problems
   * whose tc_gemm_gn_rows() is non-zero; tc_gemm_bf16 returns TC_ESHAPE otherwise. */
  float* gn_part;
} TcGemmParams;

/* C = epilogue(gather(A) * W^T), bf16 MFMA, fp32 accumulate.
 * Replaces: nn.Linear (lvdm/modules/attention.py:53-57,75-76,269,290,336,362,418,438;
 * openaimodel3d.py:170,371-380), nn.Conv2d 3x3/1x1 (openaimodel3d.py:68-70,96,154,179,187,386,545;
 * autoencoder_dualref.py:52-68,419-462,921-927), nn.Conv3d (3,1,1) (openaimodel3d.py:255-266;
 * autoencoder_dualref.py:601-605,641-648), F.interpolate nearest x2 (openaimodel3d.py:98-106),
 * GEGLU (attention.py:420-422) and the residual/embedding adds around them. */
int tc_gemm_bf16(const TcGemmParams* p, void* stream);
/* bytes of TcGemmParams.workspace this problem would use (0: it is not a split-K candidate) */
int64_t tc_gemm_workspace(const TcGemmParams* p);
/* ABI 8 -- 1 if tc_gemm_bf16 would run this problem on the weight-stationary K = 320 kernel (csrc/gemm_ws.hip), the
 * only one that accepts a_norm = 1: the host asks BEFORE it decides to drop a LayerNorm launch. */
int tc_gemm_ws_eligible(const TcGemmParams* p);
/* ABI 9 -- row-block height of TcGemmParams.gn_part for this problem under the current routing (160: the 160x160-tile
 * kernel; 128: the 128x128-tile kernel without split-K), or 0 when the kernel that would run it cannot emit
 * statistics (the caller then leaves gn_part NULL and the consumer computes its own). */
int tc_gemm_gn_rows(const TcGemmParams* p);

/* ABI 7 -- MX block-scaled fp8 GEMM (BASELINE.json configs[4]: the CDNA4 fp8 MFMA GEMM path).
 * Operands are OCP MX "MXFP8": e4m3 elements with one E8M0 (power-of-two) scale per 32 consecutive K
 * elements of a row (OCP Microscaling Formats v1.0 section 5.1/6.3: shared exponent = floor(log2(amax)) - 8,
 * elements = RNE(x / 2^shared) saturated to +-448).  The reference has no fp8 path; the tensors this
 * replaces are the same nn.Linear / nn.Conv2d / nn.Conv3d operands as tc_gemm_bf16 above, and the parity
 * oracle is oracle/mx.py (exact restatement of the quantiser + fp32 matmul of the dequantised operands). */
typedef struct TcGemmMxParams {
  TcGemmParams g;        /* as for tc_gemm_bf16, except: a / w point to fp8 e4m3 BYTES, lda / ldw / stride_a /
                            stride_w are in elements (= bytes, multiples of 16), k and cin multiples of 32 (linear)
                            / 64 (taps); c, bias, row_bias, residual and the epilogue are unchanged (bf16 / fp32);
                            workspace is unused (no split-K), batch must be 1 */
  const uint8_t* a_scale; /* E8M0 [source rows, lda_s]: byte j of a row scales its K elements [32 j, 32 j + 32) */
  const uint8_t* w_scale; /* E8M0 [n, ldw_s] */
  int32_t lda_s, ldw_s;   /* leading dimensions of the scale matrices in bytes (multiples of 4; >= ceil(k / 128) * 4
                             for w and linear a, >= cin / 32 for a convolution source) */
} TcGemmMxParams;
int tc_gemm_mxfp8(const TcGemmMxParams* p, void* stream);
/* bf16 rows [rows, ld] (first k columns, k % 32 == 0) -> e4m3 bytes q[rows, ldq] + E8M0 scales s[rows, lds];
 * scale columns [k / 32, lds) are zero-filled so a K tail of the GEMM reads finite scales. */
int tc_quant_mxfp8(const tc_bf16* x, int64_t rows, int32_t k, int32_t ld, uint8_t* q, int32_t ldq,
                   uint8_t* s, int32_t lds, void* stream);
/* tc_layernorm followed by tc_quant_mxfp8 in one pass (the quantiser fused into its producer): nn.LayerNorm
 * (attention.py:225-227) whose only consumer is an MXFP8 GEMM (qkv / GEGLU projections).  Bit-identical to the two
 * separate calls.  c % 32 == 0; lds >= ceil(c / 128) * 4 as for tc_quant_mxfp8. */
int tc_layernorm_mxfp8(const tc_bf16* x, uint8_t* q, int32_t ldq, uint8_t* s, int32_t lds, const float* gamma,
                       const float* beta, int32_t rows, int32_t c, float eps, void* stream);

/* ABI 9 -- the feed-forward of a level-0 transformer block as ONE launch (lvdm/modules/attention.py:415-442 behind
 * norm3, attention.py:244-246):  out = x + w2 . GEGLU(w1 . LN(x) + b1) + b2.  The hidden tensor ([m, hidden] bf16: 210 MB
 * at the BASELINE shape) never reaches HBM.  c = 320 and hidden = 1280 only (tc_ff_geglu_fused_eligible).
 *   x    [m, ldx] bf16: the block's input rows -- LayerNorm input AND residual;
 *   w1   [2*hidden, c] bf16 in tc_gemm_bf16's GEGLU packing (every 32 rows = 16 value rows, then their 16 gate rows):
 *        the very tensor the TC_ACT_GEGLU projection takes; with ln != 0 the LayerNorm's gamma is folded in
 *        (w1[n, :] *= gamma) and b1 carries w1 . beta, as for TcGemmParams.a_norm;
 *   b1   [2*hidden] fp32, packed like w1's rows;   w2 [c, hidden] bf16;   b2 [c] fp32;   out [m, ldo] bf16;
 *   ln   != 0: rows are normalised, (x - mean) * rsqrt(var + ln_eps) with fp32 statistics, before the first product;
 *        0: x is taken as already normalised for the product (and still added as the residual). */
typedef struct TcFfParams {
  const tc_bf16* x; const tc_bf16* w1; const float* b1; const tc_bf16* w2; const float* b2; tc_bf16* out;
  int32_t m, c, hidden, ldx, ldo, ln;
  float ln_eps;
} TcFfParams;
int tc_ff_geglu_fused_eligible(const TcFfParams* p);
int tc_ff_geglu_fused(const TcFfParams* p, void* stream);

/* ABI 9 -- the temporal self-attention of a level-0 transformer block as ONE launch (lvdm/modules/attention.py:81-144 over
 * the T = 16 frames of a pixel, called from TemporalTransformer attention.py:365-412 behind norm1 / norm2, attention.py:
 * 225-246):  out = x + wo . Attn_frames(wqkv . LN(x) + bqkv) + bo.  The [rows, 960] qkv tensor and the [rows, 320]
 * attention output never reach HBM.  c = 320, heads = 5 (of 64), t = 16, hw % 8 == 0 only (tc_temporal_attn_fused_eligible).
 *   x     [b*t*hw, ldx] bf16, row = (batch * t + frame) * hw + pixel: LayerNorm input AND residual;
 *   wqkv  [3*c, c] bf16: rows [0, c) = to_q, [c, 2c) = to_k, [2c, 3c) = to_v (head h at h*64), the fused projection
 *         tc_gemm_bf16 takes in front of tc_attn_temporal; with ln != 0 the LayerNorm's gamma is folded in and bqkv
 *         carries wqkv . beta (zeros otherwise: the reference's projections have no bias);
 *   wo    [c, c] bf16, bo [c] fp32 (to_out);   out [b*t*hw, ldo] bf16;   scale = 64^-0.5;
 *   ln    != 0: rows are normalised ((x - mean) * rsqrt(var + ln_eps), fp32 statistics) before the projection. */
typedef struct TcTbParams {
  const tc_bf16* x; const tc_bf16* wqkv; const float* bqkv; const tc_bf16* wo; const float* bo; tc_bf16* out;
  int32_t b, t, hw, c, heads, ldx, ldo, ln;
  float ln_eps, scale;
} TcTbParams;
int tc_temporal_attn_fused_eligible(const TcTbParams* p);
int tc_temporal_attn_fused(const TcTbParams* p, void* stream);

/* ABI 13 -- the fused q / k / v projection of a temporal self-attention and the attention itself as ONE launch
 * (lvdm/modules/attention.py:81-144: to_q / to_k / to_v at :96-102, the per-head softmax(q k^T * scale) v over the T = 16
 * frames of a pixel at :103-134 -- called with context = None from TemporalTransformer, attention.py:365-412):
 *     out[:, h*64 .. h*64+63] = Attn_frames(x . wqkv[q_h | k_h | v_h]^T + bqkv),     every head h
 * i.e. tc_gemm_bf16(x, wqkv) followed by tc_attn_temporal, without the [rows, 3c] tensor between them reaching HBM.
 * `out` is what to_out (a tc_gemm_bf16 with bias and residual) takes next.  t = 16, c = heads * 64, hw % 8 == 0
 * (tc_temporal_qkv_attn_eligible): every level of the UNet (c = 320 ... 1280; since round 6 level 0 takes LayerNorm -> this ->
 * to_out by default, tc_temporal_attn_fused only with TC_TB_FUSED=1).
 *   x     [b*t*hw, ldx] bf16, row = (batch * t + frame) * hw + pixel: the projection's input (the LayerNorm's output);
 *   wqkv  [3*c, c] bf16: rows [0, c) = to_q, [c, 2c) = to_k, [2c, 3c) = to_v (head h at h*64) -- tc_gemm_bf16's operand;
 *   bqkv  [3*c] fp32 or NULL (the reference's projections have no bias);   out [b*t*hw, ldo] bf16;   scale = 64^-0.5. */
typedef struct TcTqaParams {
  const tc_bf16* x; const tc_bf16* wqkv; const float* bqkv; tc_bf16* out;
  int32_t b, t, hw, c, heads, ldx, ldo;
  float scale;
} TcTqaParams;
int tc_temporal_qkv_attn_eligible(const TcTqaParams* p);
int tc_temporal_qkv_attn(const TcTqaParams* p, void* stream);

typedef struct TcAttnParams {
  const tc_bf16* q; const tc_bf16* k; const tc_bf16* v; tc_bf16* o;
  int32_t batch, heads, lq, lk;
  /* element (b, h, i, d) lives at ptr + b*sb + i*ss + h*64 + d  (head dim 64, contiguous) */
  int64_t q_sb, k_sb, v_sb, o_sb;
  int32_t q_ss, k_ss, v_ss, o_ss;
  int32_t kv_bdiv;     /* K/V batch index = b / kv_bdiv (shared reference / text keys) */
  int32_t accumulate;  /* 1: o += result (second softmax of the image cross-attention) */
  float scale;         /* d^-0.5 */
  /* ABI 6 -- optional SECOND key/value set with its own softmax, summed into the same output in one launch:
   * o = softmax(q k^T) v + softmax(q k2^T) v2 -- the text + image cross-attention of attention.py:153-207
   * (77 text keys shared by the frames of a clip, 16 image keys per frame).  k2 == NULL: single set. */
  const tc_bf16* k2; const tc_bf16* v2;
  int32_t lk2, kv2_bdiv;
  int64_t k2_sb, v2_sb;
  int32_t k2_ss, v2_ss;
} TcAttnParams;

/* softmax(q k^T * scale) v, head dim 64, flash-style (scores never leave the CU).
 * Replaces xformers.ops.memory_efficient_attention at lvdm/modules/attention.py:175,187 and
 * lvdm/models/autoencoder_dualref.py:316,326 (and the einsum fallback attention.py:103-134). */
int tc_attn_d64(const TcAttnParams* p, void* stream);

/* Temporal self-attention over <=16 frames at every pixel (attention.py:81-144 via
 * TemporalTransformer, attention.py:365-412).  qkv: fused [rows, 3*C] projection with
 * row = (b*T + t)*HW + p; columns [0,C)=q, [C,2C)=k, [2C,3C)=v, head h at h*64.
 * out: [rows, C] bf16. */
int tc_attn_temporal(const tc_bf16* qkv, tc_bf16* out, int32_t b, int32_t t, int32_t hw,
                     int32_t heads, float scale, void* stream);

/* GroupNorm(32 groups) over channels-last rows, fp32 statistics, optional fused SiLU.
 * x: [samples, rows, C]; statistics over (rows, C/32) per (sample, group): samples = B*T,
 * rows = H*W for the per-frame norms; samples = B, rows = T*H*W for the clip-wide ones.
 * Replaces GroupNormSpecific / nn.GroupNorm + nn.SiLU (lvdm/basics.py:76-87;
 * openaimodel3d.py:152-153,176-177,256-265; attention.py:265,331; autoencoder_dualref.py:29-32).
 * workspace: tc_groupnorm_workspace() bytes of fp32 partial sums. */
int64_t tc_groupnorm_workspace(int32_t samples, int32_t rows, int32_t c);
int tc_groupnorm(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                 int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                 void* workspace, int64_t workspace_bytes, void* stream);
/* ABI 9 -- the same with the statistics taken from a producer's partial sums (TcGemmParams.gn_part:
 * [samples * rows / part_rows][2][c] fp32; rows % part_rows == 0): a reduction over the partials (fp64, fixed order)
 * and ONE pass over x.  workspace: tc_groupnorm_workspace() bytes. */
int tc_groupnorm_part(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta, const float* part,
                      int32_t part_rows, int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ABI 11 removed the three ABI-10 entry points tc_groupnorm_scale_shift / tc_conv_gn_eligible / tc_conv_gn_bf16 -- GroupNorm
 * applied inside the convolution that consumes it: parity-green on the GPU and 4 % SLOWER per clip than the two operators
 * (profiles/r05_fuse_clip_ab.txt).  A cooperative single-launch GroupNorm measured in the same round
 * (x read once, blocks meeting at an atomic counter) was correct and slower on every shape as well
 * (profiles/r05_gn_coop_bench.txt) and never entered the ABI; its source is kept as scripts/experiments/gn_coop.hip.txt. */

/* LayerNorm over the last axis of [rows, C] (attention.py:225-227), eps 1e-5, affine. */
int tc_layernorm(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                 int32_t rows, int32_t c, float eps, void* stream);

/* ABI 12 -- the same two norms, with a WEIGHT PREFETCH riding on the launch.  In the reference every norm is followed by the
 * layer that consumes it (openaimodel3d.py:152-154,176-179,255-266: GroupNorm -> SiLU -> convolution; attention.py:242-246:
 * LayerNorm -> attention / feed-forward projections), and inside a forward that layer's weights are always cold: 2.9 GB of
 * parameters cycle through the 256 MB Infinity Cache once per forward.  For the 1280-channel levels of the UNet that costs
 * the GEMM 7-29 % (profiles/r05_cold_operand_probe.txt); a read of the weights one launch ahead removes 93-100 % of it
 * (profiles/r05_prefetch_premise_probe.txt).  `prefetch` names up to TC_PREFETCH_MAX read-only device buffers (ptr 16-byte
 * aligned; whole 16-byte units of `bytes` are read, never a byte beyond); extra blocks of the norm's launch stream them
 * through a load and drop the values -- y is bit-identical to tc_groupnorm / tc_layernorm, nothing else is written.
 * prefetch == NULL or n == 0: exactly the plain entry point. */
#define TC_PREFETCH_MAX 4
typedef struct TcPrefetch {
  const void* ptr[TC_PREFETCH_MAX];
  int64_t bytes[TC_PREFETCH_MAX];
  int32_t n;
} TcPrefetch;
int tc_groupnorm_pf(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                    int32_t samples, int32_t rows, int32_t c, float eps, int32_t silu,
                    void* workspace, int64_t workspace_bytes, const TcPrefetch* prefetch, void* stream);
int tc_layernorm_pf(const tc_bf16* x, tc_bf16* y, const float* gamma, const float* beta,
                    int32_t rows, int32_t c, float eps, const TcPrefetch* prefetch, void* stream);

/* Row softmax fp32 [rows, n] -> bf16 [rows, ldo] for attention computed as GEMM + softmax + GEMM: the
 * single-head d=512 mid attention of the decoder (autoencoder_dualref.py:172-200) and the OpenCLIP towers
 * (head dim 80 / causal text attention, condition.py:215-231,340-372).  Columns [n, n_out) of p are written
 * as zeros (K padding of the following P.V GEMM).  causal_period > 0: row r attends columns
 * 0 .. (r mod causal_period) only -- the text tower's attn_mask -- masked columns get probability 0.  (ABI 5) */
int tc_softmax_rows(const float* s, tc_bf16* p, int32_t rows, int32_t n, int32_t n_out, int32_t lds, int32_t ldo,
                    int32_t causal_period, void* stream);

/* (B, C, T, H, W) fp32, optionally two tensors concatenated on C, -> channels-last bf16
 * [B*T*H*W, c_pad] zero padded, times `scale`.  The hybrid-conditioning concat of
 * ddpm3d.py:1260-1264 and the `b c t h w -> (b t) c h w` rearranges (openaimodel3d.py:566). */
int tc_nchw_to_rows(const float* x0, int32_t c0, const float* x1, int32_t c1, tc_bf16* out,
                    int32_t c_pad, int32_t b, int32_t t, int32_t hw, float scale, void* stream);
/* rows -> (B, C, T, H, W) fp32: the inverse rearrange (openaimodel3d.py:602). src may be bf16 or fp32. */
int tc_rows_to_nchw(const void* src, int32_t src_f32, int32_t ld, float* out, int32_t c,
                    int32_t b, int32_t t, int32_t hw, void* stream);
/* Channel concat of two rows tensors [rows, ca] | [rows, cb] -> [rows, ca+cb]: the UNet skip
 * connections (openaimodel3d.py:596). */
int tc_concat_rows(const tc_bf16* a, int32_t ca, const tc_bf16* b, int32_t cb, tc_bf16* out,
                   int64_t rows, void* stream);

/* Sinusoidal embedding [cos | sin] (utils_diffusion.py:19-23) -> bf16 [n, dim_pad], then optional SiLU elementwise helper. */
int tc_timestep_embedding(const float* t, tc_bf16* out, int32_t n, int32_t dim, int32_t ld, void* stream);
int tc_silu_f32_to_bf16(const float* x, tc_bf16* y, int64_t n, void* stream);
/* ABI 9 -- the same from the int64 timesteps the samplers hand over (ddim.py:166 `torch.full(..., dtype=torch.long)`,
 * openaimodel3d.py:567-575): the `.to(float32)` in front of the embedding is a launch of its own otherwise. */
int tc_timestep_embedding_i64(const int64_t* t, tc_bf16* out, int32_t n, int32_t dim, int32_t ld, void* stream);

/* ABI 9 -- dst[k * rows + r, :] = src[r, :] for k < n (rows of row_bytes bytes, a multiple of 16; both contiguous):
 * where batched guidance leaves the shared prefix (lvdm/common.py: CfgShare; the reference runs the n passes as n
 * separate UNet calls, ddim.py:226-233) the single-copy rows and embedding are repeated n-fold. */
int tc_repeat_rows(const void* src, void* dst, int64_t rows, int32_t row_bytes, int32_t n, void* stream);

/* AE3DConv tail (autoencoder_dualref.py:929-935): Conv3d 3->3 (3,1,1) over T on the fp32
 * [B*T*HW, ld] rows produced by the 128->3 conv, written as (B, 3, T, H, W) fp32. */
int tc_time_mix3(const float* rows, int32_t ld, const float* w, const float* bias, float* out,
                 int32_t b, int32_t t, int32_t hw, void* stream);

/* Output path, replaces scripts/evaluation/inference.py:148-153 on the device:
 * clamp(x, -1, 1) -> (x + 1) / 2 -> (* 255) -> uint8 (truncation) with the permute (c t h w) -> (t h w c),
 * for each of the b clips of x (b, 3, t, h, w) fp32; out is (b, t, h, w, 3) uint8.  Same fp32 operations in
 * the same order as the reference, so the bytes are identical; it makes the rank-0 gather 4x smaller. */
int tc_video_to_u8(const float* x, uint8_t* out, int32_t b, int32_t t, int32_t hw, void* stream);

typedef struct TcDdimParams {
  const float* x;        /* current latent (B, n) fp32 */
  const float* e_cond;   /* UNet output for the conditional pass (B, n) */
  const float* e_uncond; /* unconditional pass, or NULL */
  const float* noise;    /* N(0,1) draw or NULL (sigma == 0) */
  float* x_prev; float* pred_x0;
  int32_t b; int64_t n;  /* n = C*T*H*W */
  float cfg_scale, guidance_rescale;
  /* fp32 scalars in the reference's order of operations (ddim.py:251-277, ddpm3d.py:240-252) */
  float sqrt_ac, sqrt_1m_ac, sqrt_a_prev, dir_coef /* sqrt(1-a_prev-sigma^2) */, sigma, x0_rescale;
  /* ABI 3 -- three-way guidance of samplers/ddim_multiplecond.py:226-236: when e_uncond_img is not NULL the
   * combination is e_uncond + cfg_img*(e_uncond_img - e_uncond) + cfg_scale*(e_cond - e_uncond_img)
   * (e_uncond_img = UNet pass with the image condition kept and the text dropped) */
  const float* e_uncond_img;
  float cfg_img;
} TcDdimParams;

/* CFG combine + rescale_noise_cfg (unbiased std over each sample) + v->(eps,x0) + dynamic
 * rescale + x_prev, fused.  Replaces ddim.py:226-277 / ddim_multiplecond.py:226-288 and
 * utils_diffusion.py:147-158.
 * workspace: b * 64 * 4 doubles. */
int64_t tc_ddim_workspace(int32_t b);
int tc_ddim_step(const TcDdimParams* p, void* workspace, int64_t workspace_bytes, void* stream);

/* introspection */
int tc_abi_version(void);
const char* tc_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* TOONCRAFTER_HIP_H */
