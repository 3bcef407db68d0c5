// Standalone A/B harness for the tc_gemm_bf16 tile families (no Python, no torch: a gpurun call spends its minutes on
// kernels, not on imports).  For every shape: the default routing with the 8-wave kernel switched off (TC_GEMM8=0) is the
// REFERENCE result; every other arm (environment per call -- the library reads its tuning switches per call) is checked
// against it element by element and timed in interleaved rounds over operand sets that rotate past the Infinity Cache.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/gemm8_bench.cpp -Ltooncrafter_amd -ltooncrafter_hip \
//         -Wl,-rpath,'$ORIGIN/../../tooncrafter_amd' -o scripts/bin/gemm8_bench
//   scripts/bin/gemm8_bench [filter-substring] > gpurun_out/gemm8_bench.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "tooncrafter_hip.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {   // uniform [-1, 1)
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((rng_state >> 40) & 0xffffff) / 8388608.0f - 1.0f;
}
static void* dev_bf16(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf(urand() * scale);
  void* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static void* dev_f32(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = urand() * scale;
  void* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

struct Arm { const char* name; std::vector<std::pair<const char*, const char*>> env; };
static const char* kSwitches[] = {"TC_GEMM8", "TC_GEMM_TILE", "TC_GEMM_TILE16", "TC_GEMM_WIDE", "TC_GEMM_WS", "TC_GEMM_SPLITK", "TC_GEMM_PIPE", "TC_G8_ABLATE", "TC_G8_STAGGER", "TC_G8_GRID", "TC_GEMM_AP", "TC_AP_GRID"};
static void set_env(const Arm& a) {
  for (const char* s : kSwitches) unsetenv(s);
  for (auto& kv : a.env) setenv(kv.first, kv.second, 1);
}

struct Shape {
  std::string tag;
  int m, n, cin;           // cin = K of a linear layer / channels per tap
  int kind;                // 0 linear, 1 3x3, 2 t3
  int frames, h, w;        // conv geometry (m = frames*h*w)
  bool geglu, res, rowbias;
};

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const int rounds = 5, iters = 12;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s | %s | %d CUs\n", tc_build_info(), prop.name, prop.multiProcessorCount);
  std::vector<Shape> shapes = {
      {"square 4096", 4096, 4096, 4096, 0, 0, 0, 0, false, false, false},
      {"square 8192", 8192, 8192, 8192, 0, 0, 0, 0, false, false, false},
      {"L0 qkv", 81920, 960, 320, 0, 0, 0, 0, false, false, false},
      {"L0 GEGLU", 81920, 2560, 320, 0, 0, 0, 0, true, false, false},
      {"L0 ff2", 81920, 320, 1280, 0, 0, 0, 0, false, true, false},
      {"L1 qkv", 20480, 1920, 640, 0, 0, 0, 0, false, false, false},
      {"L1 GEGLU", 20480, 5120, 640, 0, 0, 0, 0, true, false, false},
      {"L1 ff2", 20480, 640, 2560, 0, 0, 0, 0, false, true, false},
      {"L2 qkv", 5120, 3840, 1280, 0, 0, 0, 0, false, false, false},
      {"L2 GEGLU", 5120, 10240, 1280, 0, 0, 0, 0, true, false, false},
      {"L2 ff2", 5120, 1280, 5120, 0, 0, 0, 0, false, true, false},
      {"conv3x3 L0 320->320", 81920, 320, 320, 1, 32, 40, 64, false, true, false},
      {"conv3x3 L0 960->320", 81920, 320, 960, 1, 32, 40, 64, false, false, true},
      {"conv3x3 L1 640->640", 20480, 640, 640, 1, 32, 20, 32, false, true, false},
      {"conv3x3 L1 1920->640", 20480, 640, 1920, 1, 32, 20, 32, false, false, true},
      {"conv3x3 L2 1280->1280", 5120, 1280, 1280, 1, 32, 10, 16, false, true, false},
      {"conv3x3 L2 2560->1280", 5120, 1280, 2560, 1, 32, 10, 16, false, false, true},
      {"convT3 L0 320->320", 81920, 320, 320, 2, 32, 1, 2560, false, true, false},
      {"convT3 L1 640->640", 20480, 640, 640, 2, 32, 1, 640, false, true, false},
      {"convT3 L2 1280->1280", 5120, 1280, 1280, 2, 32, 1, 160, false, true, false},
      {"dec conv3x3 512->512 16f", 16 * 80 * 128, 512, 512, 1, 16, 80, 128, false, false, false},
      {"dec conv3x3 256->256 4f", 4 * 160 * 256, 256, 256, 1, 4, 160, 256, false, false, false},
      {"dec conv3x3 128->128 4f", 4 * 320 * 512, 128, 128, 1, 4, 320, 512, false, false, false},
      {"ragged 1000x520x392 +res", 1000, 520, 392, 0, 0, 0, 0, false, true, false},
      {"ragged conv3x3 3x17x23 64->264", 3 * 17 * 23, 264, 64, 1, 3, 17, 23, false, true, true},
      {"ragged convT3 32x70 128->136", 32 * 70, 136, 128, 2, 32, 1, 70, false, false, false},
      {"one K-tile 300x256x64", 300, 256, 64, 0, 0, 0, 0, false, false, false},
      {"two K-tiles GEGLU 512x512x128", 512, 512, 128, 0, 0, 0, 0, true, false, false},
      {"three K-tiles 256x256x192", 256, 256, 192, 0, 0, 0, 0, false, true, false},
  };
  std::vector<Arm> arms = {
      {"default(ref)", {{"TC_GEMM8", "0"}, {"TC_GEMM_AP", "0"}}},
      {"gemm8", {{"TC_GEMM8", "2"}, {"TC_GEMM_AP", "0"}}},
  };
  if (argc > 2 && !strcmp(argv[2], "stg")) {     // de-phasing sweep
    arms.push_back({"stg1", {{"TC_GEMM8", "2"}, {"TC_G8_STAGGER", "1"}}});
    arms.push_back({"stg2", {{"TC_GEMM8", "2"}, {"TC_G8_STAGGER", "2"}}});
    arms.push_back({"stg3", {{"TC_GEMM8", "2"}, {"TC_G8_STAGGER", "3"}}});
    arms.push_back({"stg4", {{"TC_GEMM8", "2"}, {"TC_G8_STAGGER", "4"}}});
    arms.push_back({"g248", {{"TC_GEMM8", "2"}, {"TC_G8_GRID", "248"}}});
    arms.push_back({"ab64", {{"TC_GEMM8", "2"}, {"TC_G8_ABLATE", "64"}}});
    arms.push_back({"ab65", {{"TC_GEMM8", "2"}, {"TC_G8_ABLATE", "65"}}});
    arms.push_back({"ab1", {{"TC_GEMM8", "2"}, {"TC_G8_ABLATE", "1"}}});
  }
  if (argc > 2 && !strcmp(argv[2], "ab")) {      // ablation arms (linear shapes only; their results are wrong by design)
    static const char* abs[] = {"1", "2", "3", "4", "6", "7", "8", "16", "32", "64", "65"};
    static std::string names[11];
    for (int i = 0; i < 11; ++i) {
      names[i] = std::string("ab") + abs[i];
      arms.push_back({names[i].c_str(), {{"TC_GEMM8", "2"}, {"TC_G8_ABLATE", abs[i]}}});
    }
  }
  for (const Shape& sh : shapes) {
    if (filter[0] && sh.tag.find(filter) == std::string::npos) continue;
    const int taps = sh.kind == 1 ? 9 : (sh.kind == 2 ? 3 : 1);
    const int k = sh.cin * taps;
    const int n_out = sh.geglu ? sh.n / 2 : sh.n;
    const size_t a_elems = (size_t)sh.m * sh.cin, c_elems = (size_t)sh.m * n_out;
    size_t per_set = (a_elems + c_elems * (sh.res ? 2 : 1)) * 2;
    int sets = (int)std::min<size_t>(6, std::max<size_t>(1, (600ull << 20) / per_set + 1));
    std::vector<void*> a(sets), r(sets), c(sets);
    for (int i = 0; i < sets; ++i) {
      a[i] = dev_bf16(a_elems, 1.0f);
      r[i] = sh.res ? dev_bf16(c_elems, 1.0f) : nullptr;
      CK(hipMalloc(&c[i], c_elems * 2));
    }
    void* w = dev_bf16((size_t)sh.n * k, 1.0f / std::sqrt((float)k) * 1.7f);
    float* bias = (float*)dev_f32(sh.n, 1.0f);
    const int row_div = sh.kind == 1 ? sh.h * sh.w : (sh.kind == 2 ? sh.w : sh.m);
    float* rb = sh.rowbias ? (float*)dev_f32((size_t)((sh.m + row_div - 1) / row_div) * sh.n, 1.0f) : nullptr;
    void* ws = nullptr;
    const int64_t ws_bytes = 512ll << 20;
    CK(hipMalloc(&ws, ws_bytes));

    auto params = [&](int set, void* out) {
      TcGemmParams p;
      memset(&p, 0, sizeof(p));
      p.a = (const tc_bf16*)a[set]; p.w = (const tc_bf16*)w; p.c = out; p.bias = bias;
      p.row_bias = rb; p.residual = (const tc_bf16*)r[set];
      p.m = sh.m; p.n = sh.n; p.k = k; p.lda = sh.cin; p.ldw = k; p.ldc = n_out; p.ldr = n_out; p.ldrb = sh.n;
      p.row_div = row_div; p.alpha = 1.f; p.out_scale = 1.f; p.act = sh.geglu ? TC_ACT_GEGLU : TC_ACT_NONE;
      p.gather = sh.kind; p.cin = sh.cin; p.batch = 1;
      if (sh.kind) { p.frames = sh.frames; p.t_len = 16; p.h_out = sh.h; p.w_out = sh.w; p.h_in = sh.h; p.w_in = sh.w; p.stride = 1; p.pad = 1; }
      p.workspace = ws; p.workspace_bytes = ws_bytes;
      return p;
    };
    // ---- correctness: every arm on set 0 against arm 0
    std::vector<std::vector<uint16_t>> outs;
    bool ok_all = true;
    std::string notes;
    for (size_t ai = 0; ai < arms.size(); ++ai) {
      set_env(arms[ai]);
      CK(hipMemset(c[0], 0xff, c_elems * 2));
      TcGemmParams p = params(0, c[0]);
      const int rc = tc_gemm_bf16(&p, nullptr);
      CK(hipDeviceSynchronize());
      if (rc != 0) { notes += std::string(" [") + arms[ai].name + " rc=" + std::to_string(rc) + "]"; outs.emplace_back(); ok_all = false; continue; }
      outs.emplace_back(c_elems);
      CK(hipMemcpy(outs.back().data(), c[0], c_elems * 2, hipMemcpyDeviceToHost));
      if (ai > 0 && !outs[0].empty()) {
        double num = 0, den = 0; float worst = 0; size_t nbad = 0;
        for (size_t i = 0; i < c_elems; ++i) {
          const float x = bf2f(outs[ai][i]), y = bf2f(outs[0][i]);
          if (!(x == x)) { ++nbad; continue; }
          num += (double)(x - y) * (x - y); den += (double)y * y;
          worst = std::max(worst, std::fabs(x - y) / (std::fabs(y) + 1.0f));
        }
        const double rel = std::sqrt(num / (den + 1e-30));
        char buf[160];
        snprintf(buf, sizeof buf, " [%s vs ref: rel-L2 %.2e, worst |d|/(|y|+1) %.2e, NaN %zu]", arms[ai].name, rel, worst, nbad);
        notes += buf;
        if (rel > 2e-3 || worst > 2e-2 || nbad) ok_all = false;   // two fp32 summation orders rounded to bf16
      }
    }
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> us(arms.size());
    for (int rd = 0; rd < rounds + 1; ++rd)
      for (size_t ai = 0; ai < arms.size(); ++ai) {
        if (outs[ai].empty()) continue;
        set_env(arms[ai]);
        CK(hipEventRecord(e0, nullptr));
        for (int it = 0; it < iters; ++it) {
          TcGemmParams p = params(it % sets, c[it % sets]);
          tc_gemm_bf16(&p, nullptr);
        }
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rd > 0) us[ai].push_back(ms * 1e3f / iters);
      }
    const double fl = 2.0 * sh.m * sh.n * (double)k;
    printf("%-34s %7dx%5dx%5d%s%s%s |", sh.tag.c_str(), sh.m, sh.n, k, sh.geglu ? " geglu" : "", sh.res ? " +res" : "", sh.rowbias ? " +rowbias" : "");
    float ref_t = 0;
    for (size_t ai = 0; ai < arms.size(); ++ai) {
      if (us[ai].empty()) { printf(" %s: --", arms[ai].name); continue; }
      std::sort(us[ai].begin(), us[ai].end());
      const float med = us[ai][us[ai].size() / 2];
      if (ai == 0) ref_t = med;
      printf(" %s %8.1f us %6.0f TF/s", arms[ai].name, med, fl / med / 1e6);
      if (ai > 0 && ref_t > 0) printf(" x%.3f", ref_t / med);
      printf(" |");
    }
    printf(" %s%s\n", ok_all ? "OK" : "MISMATCH", notes.c_str());
    fflush(stdout);
    for (int i = 0; i < sets; ++i) { CK(hipFree(a[i])); if (r[i]) CK(hipFree(r[i])); CK(hipFree(c[i])); }
    CK(hipFree(w)); CK(hipFree(bias)); if (rb) CK(hipFree(rb)); CK(hipFree(ws));
  }
  return 0;
}
