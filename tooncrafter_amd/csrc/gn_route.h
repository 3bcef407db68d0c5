// Which GroupNorm kernel takes a problem: shared by norm.hip (tc_groupnorm) and gn_coop.hip (tc_groupnorm_coop_grid), so that
// the cooperative kernel's advice ("where the single-block one-pass kernel does not apply") cannot drift from the rule itself.
#pragma once
#include <stdint.h>

struct GnOnepass {
  int u, vu, nv;          // channels per unit, 8-channel vectors per unit, vectors per thread (4 or 13); nv = 0: not taken
};

// gn_onepass_kernel (norm.hip): a (sample, unit) slab fits ONE 256-thread block's registers and the grid fills the chip
// (>= 128 blocks) or the tensor is tiny (<= 4 MiB) -- measured, profiles/r02_gn_onepass_ab.txt
static inline GnOnepass gn_onepass_rule(int samples, int rows, int c) {
  GnOnepass r{0, 0, 0};
  const int cpg = c / 32;
  if (cpg <= 0) return r;
  int u = cpg;
  while (u % 8) u += cpg;                                       // lcm(8, cpg) = U channels per unit
  const int vu = u / 8, gu = u / cpg;
  r.u = u;
  r.vu = vu;
  if ((c % u) != 0 || gu > 4 || vu > 64) return r;
  const int64_t nblk = (int64_t)(c / u) * samples;
  const int64_t bytes = (int64_t)rows * vu * 16 * samples * (c / u);
  if (!(nblk >= 128 || bytes <= (4 << 20))) return r;
  auto fits = [&](int t, int nv) { return (int64_t)((rows + t / vu - 1) / (t / vu)) <= nv; };
  r.nv = fits(256, 4) ? 4 : (fits(256, 13) ? 13 : 0);
  return r;
}
