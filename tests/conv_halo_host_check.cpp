// Host-side execution of the index arithmetic of csrc/conv_halo.hip: THE SAME functions the kernel calls
// (csrc/conv_halo_index.h, namespace chx) drive a lane-by-lane model of one block -- halo vectors through a byte-addressed
// LDS image, W tiles as the LDS-DMA would lay them down, v_mfma_f32_16x16x32_bf16 operand / result layouts, the K loop's
// (chunk, tap) order, the K split over two groups, the epilogue's row map -- and the result is compared with a direct
// convolution on small integers (every sum exact).  Built with g++ by tests/test_conv_halo_host_cpu.py; no GPU, no HIP.
// Reading an LDS unit nobody wrote is an error (NaN poison), as is a halo vector outside its buffer.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_halo_index.h"

using namespace chx;

struct Rng {
  uint64_t s;
  int next(int lo, int hi) {            // [lo, hi]
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return lo + (int)((s >> 33) % (uint64_t)(hi - lo + 1));
  }
};

typedef std::vector<double> Vec;

template <int GATHER, int WM, int KS>
static int run(int frames, int h, int w, int cin, int n, int lda, const char* tag) {
  using S = Shape<GATHER, WM>;
  const int taps = S::TAPS, hw = h * w, m = frames * hw, K = taps * cin;
  Rng rng{12345};
  Vec x((size_t)m * lda), wt((size_t)n * K), out((size_t)m * n, NAN), ref((size_t)m * n, 0.0);
  for (auto& v : x) v = rng.next(-3, 3);
  for (auto& v : wt) v = rng.next(-2, 2);
  const Vec& xn = x;                                               // what the convolution sees
  // ---- direct convolution (weights tap-major: K = tap * cin + c; 3x3 tap = 3 (dy + 1) + (dx + 1); temporal tap = dt + 1)
  for (int f = 0; f < frames; ++f)
    for (int y = 0; y < h; ++y)
      for (int xx = 0; xx < w; ++xx) {
        const int64_t mo = ((int64_t)f * h + y) * w + xx;
        for (int t = 0; t < taps; ++t) {
          int64_t src;
          if (GATHER == GATHER_3x3) {
            const int iy = y + t / 3 - 1, ix = xx + t % 3 - 1;
            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
            src = ((int64_t)f * h + iy) * w + ix;
          } else {
            const int ft = f % 16 + t - 1;
            if (ft < 0 || ft >= 16) continue;
            src = ((int64_t)(f / 16 * 16 + ft) * h + y) * w + xx;
          }
          for (int nn = 0; nn < n; ++nn) {
            double a = 0;
            for (int c = 0; c < cin; ++c) a += xn[(size_t)src * lda + c] * wt[(size_t)nn * K + t * cin + c];
            ref[(size_t)mo * n + nn] += a;
          }
        }
      }
  // ---- the kernel, block by block
  const int tm = tiles_m<GATHER, WM>(frames, h, w), tn = n / BN;
  if ((int64_t)tm * 80 * WM != m) { printf("%s: the patches do not tile the problem\n", tag); return 1; }
  const int nwaves = 2 * WM;
  for (int tile_m = 0; tile_m < tm; ++tile_m)
    for (int tile_n = 0; tile_n < tn; ++tile_n) {
      const Patch pt = patch_of<GATHER, WM>(tile_m, h, w);
      Vec total((size_t)nwaves * NT * NT * 64 * 4, 0.0);
      for (int grp = 0; grp < KS; ++grp) {
        Vec sA((size_t)S::A_BYTES / 16 * 8, NAN), sW[2] = {Vec((size_t)W_STAGE / 16 * 8, NAN), Vec((size_t)W_STAGE / 16 * 8, NAN)};
        Vec acc((size_t)nwaves * NT * NT * 64 * 4, 0.0);
        auto request_w = [&](int k0, int stage) {
          for (auto& v : sW[stage]) v = NAN;
          for (int tid = 0; tid < S::THREADS; ++tid) {
            const int lrow = w_lrow(tid), wchunk = w_chunk(tid), wave = tid >> 6, lane = tid & 63;
            for (int i = 0; i < S::RB; ++i) {
              if (!w_pass_live<WM>(i, wave)) continue;
              const int nl = lrow + S::RSTEP * i, nn = tile_n * BN + nl;
              const int dst = wave * 1024 + i * S::PIECE + lane * 16;         // the DMA lays a wave's 64 x 16 bytes down lane-linearly
              if (dst + 16 > W_STAGE) { printf("%s: W piece outside its stage\n", tag); exit(1); }
              for (int e = 0; e < 8; ++e)
                sW[stage][(size_t)dst / 16 * 8 + e] = (nl < BN && nn < n) ? wt[(size_t)nn * K + k0 + wchunk * 8 + e] : 0.0;
            }
          }
        };
        auto fill_halo = [&](int chunk) {
          for (int tid = 0; tid < S::THREADS; ++tid)
            for (int i = 0; i < NV; ++i) {
              const HaloVec hv = halo_vec<GATHER, WM>(pt, tid, i, h, w, lda);
              if (hv.lds < 0) continue;
              if (hv.lds % 16 || hv.lds + 16 > S::A_BYTES) { printf("%s: halo vector outside its buffer\n", tag); exit(1); }
              for (int e = 0; e < 8; ++e) {
                double v = 0.0;
                if (hv.off != OOB) {
                  const int64_t byte = pt.row_lo * lda * 2 + hv.off + (int64_t)chunk * (BK * 2) + e * 2;
                  if (byte / 2 >= (int64_t)m * lda) { printf("%s: halo source outside the tensor\n", tag); exit(1); }
                  v = x[(size_t)(byte / 2)];
                }
                sA[(size_t)hv.lds / 16 * 8 + e] = v;
              }
            }
        };
        auto compute = [&](int stage, int shift) {
          for (int wave = 0; wave < nwaves; ++wave) {
            const int wm = wave >> 1, wn = wave & 1;
            for (int ks = 0; ks < 2; ++ks) {
              double A[NT][16][32], B[NT][16][32];
              for (int i = 0; i < NT; ++i)
                for (int lane = 0; lane < 64; ++lane) {
                  const int frow = lane & 15, fq = lane >> 4;
                  const int a_addr = frag_a_addr(frag_a_hp0(wm, i, frow), shift, fq) ^ (ks << 6);
                  const int b_addr = frag_b_off(wn, i, frow) + frag_b_chunk(wn, frow, fq, ks);
                  if (a_addr + 16 > S::A_BYTES || b_addr + 16 > W_STAGE) { printf("%s: fragment outside its buffer\n", tag); exit(1); }
                  for (int e = 0; e < 8; ++e) {
                    A[i][frow][8 * fq + e] = sA[(size_t)a_addr / 16 * 8 + e];
                    B[i][frow][8 * fq + e] = sW[stage][(size_t)b_addr / 16 * 8 + e];
                  }
                }
              for (int i = 0; i < NT; ++i)
                for (int j = 0; j < NT; ++j)
                  for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {                             // result: col = lane & 15, row = 4 (lane >> 4) + r
                      const int row = 4 * (lane >> 4) + r, col = lane & 15;
                      double d = 0;
                      for (int k = 0; k < 32; ++k) d += A[i][row][k] * B[j][col][k];
                      acc[((((size_t)wave * NT + i) * NT + j) * 64 + lane) * 4 + r] += d;
                    }
            }
          }
        };
        // the K loop of the kernel (conv_halo.hip), statement by statement
        const int nch = (cin / BK) / KS, c0 = grp * nch, nk = S::TAPS * nch;
        request_w(w_k0(0, c0, cin), 0);
        fill_halo(c0);
        int c = 0, tap = 0, ty = 0, tx = 0;
        for (int kb = 0; kb < nk; ++kb) {
          const int st = kb & 1;
          int ntap = tap + 1, nc = c, nty = ty, ntx = tx + 1;
          if (ntx == 3) { ntx = 0; nty = ty + 1; }
          if (ntap == S::TAPS) { ntap = 0; nc = c + 1; nty = 0; ntx = 0; }
          const bool more = kb + 1 < nk;
          const bool refill = more && ntap == 0;
          if (more) request_w(w_k0(ntap, c0 + nc, cin), st ^ 1);
          compute(st, tap_shift(GATHER, ty, tx));
          if (refill) fill_halo(c0 + nc);
          tap = ntap; c = nc; ty = nty; tx = ntx;
        }
        for (size_t q = 0; q < acc.size(); ++q) total[q] += acc[q];
      }
      for (int wave = 0; wave < nwaves; ++wave) {
        const int wm = wave >> 1, wn = wave & 1;
        for (int i = 0; i < NT; ++i)
          for (int j = 0; j < NT; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int r = 0; r < 4; ++r) {
                const int frow = lane & 15, fq = lane >> 4;
                const int64_t mo = out_row(pt, wm, i, fq * 4 + r);
                const int col = tile_n * BN + wn * WT + j * 16 + frow;
                if (mo < 0 || mo >= m) { printf("%s: output row outside the problem\n", tag); exit(1); }
                double& o = out[(size_t)mo * n + col];
                if (!std::isnan(o)) { printf("%s: output (%lld, %d) written twice\n", tag, (long long)mo, col); exit(1); }
                o = total[((((size_t)wave * NT + i) * NT + j) * 64 + lane) * 4 + r];
              }
      }
    }
  size_t bad = 0;
  for (size_t q = 0; q < out.size(); ++q) bad += !(out[q] == ref[q]);
  printf("%-44s %s (%zu of %zu outputs differ)\n", tag, bad ? "FAIL" : "ok", bad, out.size());
  return bad != 0;
}

int main() {
  int rc = 0;
  rc |= run<GATHER_3x3, 2, 1>(2, 20, 32, 128, 160, 128, "3x3 2x20x32 128->160");
  rc |= run<GATHER_3x3, 2, 1>(1, 10, 16, 64, 320, 192, "3x3 1x10x16 64->320 lda 192");
  rc |= run<GATHER_3x3, 4, 1>(1, 40, 32, 64, 160, 64, "3x3 tall 1x40x32 64->160");
  rc |= run<GATHER_3x3, 2, 2>(1, 10, 32, 256, 160, 256, "3x3 K split 1x10x32 256->160");
  rc |= run<GATHER_T3, 2, 1>(16, 4, 5, 128, 160, 128, "t3 16x(4x5) 128->160");
  rc |= run<GATHER_T3, 2, 1>(32, 2, 5, 64, 160, 96, "t3 32x(2x5) 64->160 lda 96");
  rc |= run<GATHER_T3, 4, 1>(16, 8, 5, 64, 160, 64, "t3 tall 16x(8x5) 64->160");
  rc |= run<GATHER_T3, 2, 2>(16, 2, 5, 256, 160, 256, "t3 K split 16x(2x5) 256->160");
  return rc;
}
