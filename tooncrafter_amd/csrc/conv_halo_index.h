// Index arithmetic of the tap-reuse convolution kernel (conv_halo.hip), as plain functions that compile for the device AND for
// the host: the kernel calls them, and tests/conv_halo_host_check.cpp -- built with g++, no GPU -- calls THE SAME functions to
// move numbers through a byte-addressed LDS model lane by lane and compares with a direct convolution.  (The kernel was
// written in a session without GPU access; a Python restatement of its formulas, tests/test_conv_halo_cpu.py, cannot catch a
// formula that was mis-typed in the kernel only.  This header is the formulas.)
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define CHX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define CHX_HD inline
#endif

namespace chx {

constexpr int GATHER_3x3 = 1, GATHER_T3 = 2;                // = TC_GATHER_CONV3x3 / TC_GATHER_CONVT3 (static_assert in the kernel)
constexpr int HX = 18;                                      // halo patch width (16 + 2) for both geometries
constexpr int BN = 160, WT = 80, NT = 5, BK = 64;
constexpr int W_STAGE = BN * BK * 2;                        // 20 KiB
constexpr int NV = 7;                                       // halo vectors per thread and chunk
constexpr uint32_t OOB = 0x80000000u;                       // = TC_OOB: an offset no descriptor covers -> the load returns zeros

template <int GATHER, int WM>
struct Shape {                                              // 3x3: y = image row, x = pixel; temporal: y = pixel, x = frame
  static constexpr int TAPS = GATHER == GATHER_3x3 ? 9 : 3;
  static constexpr int PY = 5 * WM;                         // patch rows: 10 | 20
  static constexpr int HY = GATHER == GATHER_3x3 ? PY + 2 : PY;
  static constexpr int NPIX = HY * HX;                      // 216 | 180 | 396 | 360
  static constexpr int THREADS = 128 * WM;                  // threads of one wave group
  static constexpr int A_BYTES = (NPIX * 128 + 1023) / 1024 * 1024;
  static constexpr int RSTEP = 16 * WM;                     // W rows per loader pass: 32 | 64
  static constexpr int RB = (BN + RSTEP - 1) / RSTEP;       // passes: 5 | 3
  static constexpr int PIECE = RSTEP * BK * 2;              // LDS bytes from one pass to the next
  static_assert(NPIX * 8 <= NV * THREADS, "halo vectors per thread");
};

// ---- block -> patch
struct Patch {
  int img, pin;          // frame (3x3) | clip (temporal); patch index inside it
  int Y0, X0;            // 3x3: first image row / pixel of the patch
  int ys, xs;            // output row of patch position (y, x): m00 + y ys + x xs
  int64_t m00, row_lo;   // first output row; lowest SOURCE row the patch can touch (the SRD of A starts there)
};

template <int GATHER, int WM>
CHX_HD int patches_per_image(int h, int w) {
  return GATHER == GATHER_3x3 ? (h / Shape<GATHER, WM>::PY) * (w / 16) : (h * w) / Shape<GATHER, WM>::PY;
}
template <int GATHER, int WM>
CHX_HD int tiles_m(int frames, int h, int w) {
  return (GATHER == GATHER_3x3 ? frames : frames / 16) * patches_per_image<GATHER, WM>(h, w);
}
template <int GATHER, int WM>
CHX_HD Patch patch_of(int tile_m, int h, int w) {
  constexpr int PY = Shape<GATHER, WM>::PY;
  const int per_img = patches_per_image<GATHER, WM>(h, w);
  const int hw = h * w;
  Patch t;
  t.img = tile_m / per_img;
  t.pin = tile_m - t.img * per_img;
  t.Y0 = t.X0 = 0;
  if (GATHER == GATHER_3x3) {
    const int tpx = w / 16;
    const int ty0 = t.pin / tpx;
    t.Y0 = ty0 * PY;
    t.X0 = (t.pin - ty0 * tpx) * 16;
    t.m00 = ((int64_t)t.img * h + t.Y0) * w + t.X0;
    t.ys = w;
    t.xs = 1;
    t.row_lo = t.m00 - w - 1;
  } else {
    t.m00 = (int64_t)t.img * 16 * hw + t.pin * PY;
    t.ys = 1;
    t.xs = hw;
    t.row_lo = t.m00;
  }
  if (t.row_lo < 0) t.row_lo = 0;
  return t;
}

// ---- halo vectors of a thread: v = tid + THREADS i -> halo pixel, 16-byte segment v & 7 (8 lanes = one pixel's 128 bytes).
// Consecutive lane octets walk the direction in which SOURCE rows are adjacent -- along x for the 3x3 patch (q = 18 hy + hx),
// along the PIXELS of a frame for the temporal one (q = PY hx + hy).  The LDS image is hp = 18 hy + hx either way, its 16-byte
// segments XOR-swizzled by hp & 7.
struct HaloVec {
  uint32_t off;          // byte offset of the vector from row_lo (chunk 0) or OOB: outside the image / no such pixel
  int lds;               // byte address in the halo buffer, -1: no such pixel
};
template <int GATHER, int WM>
CHX_HD HaloVec halo_vec(const Patch& t, int tid, int i, int h, int w, int lda) {
  using S = Shape<GATHER, WM>;
  const int v = tid + S::THREADS * i;
  const int q = v >> 3, seg = v & 7;
  int hy, hx;
  if (GATHER == GATHER_3x3) { hy = q / HX; hx = q - hy * HX; }
  else { hx = q / S::PY; hy = q - hx * S::PY; }
  bool ok = q < S::NPIX;
  int64_t src;
  if (GATHER == GATHER_3x3) {
    const int iy = t.Y0 + hy - 1, ix = t.X0 + hx - 1;
    ok = ok && iy >= 0 && iy < h && ix >= 0 && ix < w;
    src = ((int64_t)t.img * h + iy) * w + ix;
  } else {
    ok = ok && hx >= 1 && hx <= 16;
    src = ((int64_t)t.img * 16 + (hx - 1)) * (h * w) + t.pin * S::PY + hy;
  }
  HaloVec r;
  r.off = ok ? (uint32_t)((src - t.row_lo) * lda * 2 + seg * 16) : OOB;
  const int pix = hy * HX + hx;
  r.lds = q < S::NPIX ? pix * 128 + ((seg ^ (pix & 7)) << 4) : -1;
  return r;
}

// ---- MFMA fragments (v_mfma_f32_16x16x32_bf16: lane holds row frow = lane & 15, k = 8 fq .. +7 of the 32-deep slice, fq = lane >> 4)
// A: halo pixel of patch position (y = 5 wm + i, x = frow) shifted by the tap; K-slice ks in {0, 1} = segments 4 ks + fq
CHX_HD int tap_shift(int gather, int ty, int tx) { return gather == GATHER_3x3 ? ty * HX + tx : tx; }
CHX_HD int frag_a_hp0(int wm, int i, int frow) { return (wm * 5 + i) * HX + frow; }
CHX_HD int frag_a_addr(int hp0, int shift, int fq) {          // K-slice 0; slice 1 = this ^ 64
  const int hp = hp0 + shift;
  return (hp << 7) + ((fq ^ (hp & 7)) << 4);
}
// W: stage image [160 rows][128 B], chunk swizzled by (row >> 1) & 7 (gemm16.hip)
CHX_HD int frag_b_off(int wn, int j, int frow) { return (wn * WT + j * 16 + frow) * (BK * 2); }
CHX_HD int frag_b_chunk(int wn, int frow, int fq, int ks) { return ((ks * 4 + fq) ^ (((wn * WT + frow) >> 1) & 7)) << 4; }

// ---- W tile requests: thread -> (row lrow + RSTEP i, SOURCE chunk); a wave's piece lands lane-linearly at wave * 1 KiB + i * PIECE
CHX_HD int w_lrow(int tid) { return tid >> 3; }
CHX_HD int w_chunk(int tid) { return (tid & 7) ^ ((w_lrow(tid) >> 1) & 7); }
template <int WM>
CHX_HD bool w_pass_live(int i, int wave) {                   // the tall block's third pass has rows for waves 0..3 only
  return WM == 2 || i < Shape<GATHER_3x3, WM>::RB - 1 || wave < 4;
}

// ---- epilogue: tile row 16 y + x (y = 5 wm + i, x = lr) -> output row
CHX_HD int64_t out_row(const Patch& t, int wm, int i, int lr) { return t.m00 + (int64_t)(wm * 5 + i) * t.ys + (int64_t)lr * t.xs; }

// ---- K loop order: step kb of a group = (chunk c0 + c, tap); W columns of the step
CHX_HD int w_k0(int tap, int chunk, int cin) { return tap * cin + chunk * BK; }

}  // namespace chx
