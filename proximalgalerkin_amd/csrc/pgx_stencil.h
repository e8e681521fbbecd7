// Helpers shared by the stencil-multigrid translation units (pgx_kernels.hip: fp64 kernels; pgx_mg32.hip: the single-precision
// V-cycle): XCD-aware block order, tile enumeration of the row-mapped kernels, kernel-argument stencil constants.
#pragma once
#include <algorithm>

#include "pgx_internal.h"

// XCD-aware block remap (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroups are dealt round-robin,
// so blocks b and b+8 share an XCD).  Logical block = the b-th block of a CONTIGUOUS range owned by one
// XCD: neighbouring rows (which re-read the same x / stencil lines) then hit the same L2 instead of
// pulling every line into up to 8 L2s.  Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ int xcd_block(int b, int nb, int enable) {
  if (!enable) return b;
  const int xcd = b & 7, k = b >> 3;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Tiles whose image is interior: tx in [1, nfx], ty in [1, nfy]; all others are "boundary tiles" (k_st_smoothRb).
struct RowmapGrid {
  int ntx, nty, nfx, nfy;
};
template <int TY, int K>
static inline RowmapGrid rowmap_grid(int nx, int ny, int fast_ok) {
  constexpr int W = 64, TX = W - 2 * K, H0 = TY + 2 * K;
  // the tile enumeration takes tx, ty >= 1 as "the image starts inside the grid": tx TX - K >= 1 and ty TY - K >= 1 at tx = ty = 1
  static_assert(TX - K >= 1 && TY - K >= 1, "first interior tile would read outside the grid");
  RowmapGrid g;
  g.ntx = (nx + TX) / TX;
  g.nty = (ny + TY) / TY;
  // fast <=> tx*TX - K >= 1, tx*TX - K + W - 1 <= nx - 1, ty*TY - K >= 1, ty*TY - K + H0 - 1 <= ny - 1
  g.nfx = (nx - 1 - (W - 1 - K)) >= TX ? (nx - 1 - (W - 1 - K)) / TX : 0;
  g.nfy = (ny - 1 - (H0 - 1 - K)) >= TY ? (ny - 1 - (H0 - 1 - K)) / TY : 0;
  g.nfx = std::min(g.nfx, g.ntx - 1);
  g.nfy = std::min(g.nfy, g.nty - 1);
  if (!fast_ok || g.nfx <= 0 || g.nfy <= 0) g.nfx = g.nfy = 0;
  return g;
}

struct RrGrid {
  int ntx, nty, nfx, nfy;  // coarse tiles; interior ("fast") tiles are tx in [1, nfx], ty in [1, nfy]
};


static inline StConst make_stconst(const GridLevel& L) {
  StConst sc;
  for (int s = 0; s < 7; ++s) {
    sc.K[s] = L.Kc[s];
    sc.M[s] = L.Mc[s];
  }
  sc.uniform = L.uniform;
  return sc;
}

