// pgx_nd_gemm.h - the dense building blocks of pgx_nd.hip that a standalone harness (tools/native/nd_gemm_bench.hip) compiles too:
// tile constants, the parent-centric gather context and the C -= A B kernels on the fp64 matrix cores.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define ND_NB 64   // widest pivot panel (diagonal block kept in LDS; one wavefront lane per row)
#define ND_TS 64   // panel chunk
#define ND_KC 16   // GEMM k-chunk staged in LDS
#define ND_SLAB 256 // pivots per triangular-solve launch in the solve phase
#define ND_OUTER 256 // (historical default of the outer block; pgx_nd::outer decides, PGX_ND_OUTER)
typedef double nd_v4d __attribute__((ext_vector_type(4)));

// Parent-centric assembly fused into the Schur update (GATHER variants of the GEMM kernels): the border block of a front is never
// written by k_nd_gather and read back - the update computes C = (children's Schur entries through the inverse maps) - A B.
struct NdGatherCtx {
  int64_t f0;  // first front of the batch
  const int32_t *child0, *child1, *fM, *fP, *inv0, *inv1;
  const int64_t *fbase, *vbase;
  int sym;  // symmetric mode (pgx_nd_set_symmetric): only the LOWER triangle of a child's Schur block is valid - entry (row, col) is
            // read at (max, min)
};
struct NdGatherSrc {
  const double *S0, *S1;
  const int32_t *I0, *I1;
  int M0, M1;
};
__device__ __forceinline__ NdGatherSrc nd_gather_src(const NdGatherCtx& g, const double* arena, int64_t f) {
  NdGatherSrc q;
  const int c0 = g.child0[f], c1 = g.child1[f];
  q.I0 = g.inv0 + g.vbase[f];
  q.I1 = g.inv1 + g.vbase[f];
  q.M0 = q.M1 = 0;
  q.S0 = q.S1 = nullptr;
  if (c0 >= 0) q.M0 = g.fM[c0], q.S0 = arena + g.fbase[c0] + (int64_t)g.fP[c0] * q.M0 + g.fP[c0];
  if (c1 >= 0) q.M1 = g.fM[c1], q.S1 = arena + g.fbase[c1] + (int64_t)g.fP[c1] * q.M1 + g.fP[c1];
  return q;
}

// broadcast of one lane's double to the wave when the lane index is wave-uniform: two v_readlane_b32 (scalar result) instead
// of the two ds_bpermute_b32 round trips through the LDS crossbar that __shfl compiles to - these broadcasts sit on the
// serial chains of the diagonal-block LU and of the triangular solves
__device__ __forceinline__ double nd_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}


// LU without pivoting of an nb x nb diagonal block (nb <= 64) by the 256 threads of a workgroup: F = the block in the working matrix
// (leading dimension M), S = its place in the compact store, D = 64 x 65 doubles of LDS.  Blocked by 8 columns:
// (a) wave 0 factors the 8-column panel in registers (lane = row, pivot rows broadcast by v_readlane, no barrier),
// (b) the 8 x rest block row of U by forward substitution, one thread per column (the coefficients are LDS broadcasts),
// (c) the rank-8 update of the trailing block on the MATRIX CORES (round 5): its 16 x 16 tiles dealt to the four waves, the C tile
//     loaded into the accumulator from LDS, two v_mfma_f64_16x16x4_f64 with the negated L tile as A operand, stored back.
// Three barriers per 8 columns.  This launch is the one purely serial link of the diag -> panel -> update chain of a level near the
// root; tools/native/nd_diag_bench.hip times it alone (s_memtime): the round-2 form took 72 000 cycles = 31 us for 64 pivots, of
// which ~700 cycles PER PIVOT in (a) - an IEEE fp64 division per lane and pivot (~200 cycles) and two v_readlane per broadcast value -
// and the rest in (c)'s 17 LDS reads per updated entry.  Now: the multiplier is a * (1 / pivot) with the reciprocal from v_rcp_f64 + two
// Newton steps (the last bits of L differ from a true division's by <= 1 ulp; the factorisation stays backward stable and bitwise
// reproducible), the index arithmetic of the load / store phases has no integer division, and (c) is 2 MFMAs per tile.
__device__ __forceinline__ double nd_recip(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = fma(fma(-p, r, 1.0), r, r);
  return fma(fma(-p, r, 1.0), r, r);
}
__device__ __forceinline__ void nd_diag_lu(double (*D)[ND_NB + 1], const double* __restrict__ F, double* __restrict__ S, int M, int nb,
                                           int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // all loads in flight before the first LDS write: thread (lane, wave) takes rows lane of the columns wave, wave + 4, ...
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = wave + 4 * q;
      v[q] = (lane < nb && c < nb) ? F[(int64_t)c * M + lane] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) D[lane][wave + 4 * q] = v[q];  // (zeros beyond nb: the MFMA tiles below read whole tiles)
  }
  __syncthreads();
  for (int jb = 0; jb < nb; jb += 8) {
    const int w = min(8, nb - jb);
    if (wave == 0) {
      const int r = lane;
      const bool act = r >= jb && r < nb;
      double a[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = (act && c < w) ? D[r][jb + c] : 0.0;
      int bad = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < w) {  // uniform
          const int pr = jb + c;
          if (r == pr && fabs(a[c]) < 1e-300) {  // exact / denormal zero pivot: static perturbation, reported via info
            a[c] = a[c] < 0 ? -1e-300 : 1e-300;
            bad = 1;
          }
          double pv[8];
#pragma unroll
          for (int c2 = 0; c2 < 8; ++c2) pv[c2] = c2 >= c ? nd_bcast(a[c2], pr) : 0.0;
          const double rp = nd_recip(pv[c]);
          if (act && r > pr) {
            const double l = a[c] * rp;
            a[c] = l;
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2)
              if (c2 > c) a[c2] -= l * pv[c2];
          }
        }
      }
      if (bad) atomicAdd(info, 1);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (act && c < w) D[r][jb + c] = a[c];
    }
    __syncthreads();
    const int rest = nb - jb - w;
    if (rest > 0) {  // (then w == 8)
      if (tid < rest) {  // U12 = L11^{-1} A12, one column per thread
        const int c = jb + 8 + tid;
        double u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          double v = D[jb + i][c];
#pragma unroll
          for (int m = 0; m < 8; ++m)
            if (m < i) v -= D[jb + i][jb + m] * u[m];
          u[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) D[jb + i][c] = u[i];
      }
      __syncthreads();
      // trailing block [jb + 8, nb)^2 -= L U in 16 x 16 tiles.  v_mfma_f64_16x16x4_f64: A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15],
      // C / D[m = (l >> 4) + 4 reg][n = l & 15].  A tile may reach beyond nb (zeros there, see the load phase) or start before
      // jb + 8 is a multiple of 16 - tiles are anchored at jb + 8 -, never beyond row / column 63 + 16: clamped reads, guarded stores.
      const int nt = (rest + 15) / 16, n = lane & 15, g = lane >> 4;
      for (int t = wave; t < nt * nt; t += 4) {
        const int R0 = jb + 8 + 16 * (t % nt), C0 = jb + 8 + 16 * (t / nt);
        const int rn = min(R0 + n, ND_NB - 1), cn = min(C0 + n, ND_NB - 1);
        nd_v4d acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = D[min(R0 + g + 4 * q, ND_NB - 1)][cn];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double lf = -D[rn][jb + 4 * q + g];
          const double uf = D[jb + 4 * q + g][cn];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lf, uf, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (R0 + g + 4 * q < nb && C0 + n < nb) D[R0 + g + 4 * q][C0 + n] = acc[q];
      }
      __syncthreads();
    }
  }
  // L11\\U11 is final: it goes to the compact factor store, where the panel solves (and the solve phase) read it
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = wave + 4 * q;
    if (lane < nb && c < nb) S[(int64_t)c * M + lane] = D[lane][c];
  }
}

// Epilogue of the GATHER Schur updates: C(i, j) = S0[inv0[j], inv0[i]] + S1[inv1[j], inv1[i]] - acc for the wave's TJ x TI MFMA tiles
// (lane l: row i = ri0 + 16 ti + (l & 15), column j = cj0 + 16 tj + (l >> 4) + 4 reg).  One implementation for both tile sizes
// (round 5), branch-free: every index is formed first (absent child / out-of-range / unmapped position -> a dummy read of entry 0,
// masked afterwards), the gathers of a (tj, reg) column go out together and all column maps are fetched before the first gather.
// Measured: no faster than the round-2 form with an `if` around each load, and neither is a variant with all 32 gathers of a 64 x 64
// tile in flight at once - these launches (2.1 TB/s of gathered + written bytes on the mid levels, where K = P <= 64 makes them
// nearly pure gathers) are not bound by how many loads a lane has outstanding; 8-byte accesses in runs of 16 lanes are what the
// MFMA accumulator layout offers the memory system.
template <int TJ, int TI>
__device__ __forceinline__ void nd_gather_epilogue(const NdGatherCtx& gc, const double* __restrict__ arena, int64_t f, double* __restrict__ F,
                                                   int M, int ri0, int cj0, int rmax, int cmax, int l, const nd_v4d (&acc)[TJ][TI]) {
  const NdGatherSrc g = nd_gather_src(gc, arena, f);
  const double* const S0 = g.S0 ? g.S0 : arena;  // (a valid address for the masked dummy reads)
  const double* const S1 = g.S1 ? g.S1 : arena;
  int a0[TI], a1[TI];  // the lane's rows in the children's borders
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int i = ri0 + 16 * ti + (l & 15);
    a0[ti] = (g.S0 && i < rmax) ? g.I0[i] : -1;
    a1[ti] = (g.S1 && i < rmax) ? g.I1[i] : -1;
  }
  int b0[TJ][4], b1[TJ][4];  // ... and its columns
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = cj0 + 16 * tj + (l >> 4) + 4 * reg;
      b0[tj][reg] = (g.S0 && j < cmax) ? g.I0[j] : -1;
      b1[tj][reg] = (g.S1 && j < cmax) ? g.I1[j] : -1;
    }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = cj0 + 16 * tj + (l >> 4) + 4 * reg;
      const int64_t c0o = (int64_t)b0[tj][reg] * g.M0, c1o = (int64_t)b1[tj][reg] * g.M1;
      double v0[TI], v1[TI];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        // (row a, column b) of the child's block sits at b * M + a; in symmetric mode above the diagonal at a * M + b
        const int64_t p0 = (gc.sym && a0[ti] < b0[tj][reg]) ? (int64_t)a0[ti] * g.M0 + b0[tj][reg] : c0o + a0[ti];
        const int64_t p1 = (gc.sym && a1[ti] < b1[tj][reg]) ? (int64_t)a1[ti] * g.M1 + b1[tj][reg] : c1o + a1[ti];
        v0[ti] = S0[(a0[ti] | b0[tj][reg]) >= 0 ? p0 : 0];
        v1[ti] = S1[(a1[ti] | b1[tj][reg]) >= 0 ? p1 : 0];
      }
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        const int i = ri0 + 16 * ti + (l & 15);
        double v = (a0[ti] | b0[tj][reg]) >= 0 ? v0[ti] : 0.0;
        if ((a1[ti] | b1[tj][reg]) >= 0) v += v1[ti];
        // (symmetric mode: an entry above the diagonal of the Schur block has no reader - not written)
        if (i < rmax && j < cmax && !(gc.sym && j > i)) F[(int64_t)j * M + i] = v - acc[tj][ti][reg];
      }
    }
}

// Which tile a workgroup computes (round 5, `order` = 1; 0 = the plain blockIdx mapping of rounds 2-4).  The dispatcher hands
// consecutive workgroups (x fastest, then y, z) to the eight XCDs in turn, each with a private 4 MB L2: with the plain mapping the
// tiles of one front were `count` workgroups apart, never resident together, so every tile fetched its L and U panels from HBM / MALL
// (4.2 MB per (128 + 512) front instead of the 1 MB the panels hold), and near the root one XCD walked down a tile COLUMN of one front,
// re-reading all its L panels per column.  Now XCD c = n mod 8 owns a CONTIGUOUS run of (front, tile) pairs - the fronts one after the
// other, the tiles of a front in groups of 8 tile rows walked column by column, so that the ~64 tiles resident on an XCD form an
// 8 x 8 block sharing 8 L and 8 U panels.  The runs are balanced to the number of workgroups each XCD really receives, which makes
// n -> (front, tile) a bijection.
__device__ __forceinline__ bool nd_block_tile(int order, int nt, int& front, int& t) {
  if (!order) {
    front = (int)blockIdx.x;
    t = -1;
    return true;
  }
  const int64_t n = (int64_t)blockIdx.x + (int64_t)gridDim.x * ((int64_t)blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
  const int64_t T = (int64_t)gridDim.x * nt, q = T >> 3, r = T & 7;
  const int c = (int)(n & 7);
  const int64_t m = c * q + (c < r ? c : r) + (n >> 3);
  front = (int)(m / nt);
  t = (int)(m - (int64_t)front * nt);
  return true;
}
// tile t of an nr x nc rectangle of tiles in groups of 8 rows, column by column inside a group
__device__ __forceinline__ void nd_grouped(int t, int nr, int nc, int& tr, int& tc) {
  const int g = t / (8 * nc), gr = min(8, nr - 8 * g), w = t - g * 8 * nc;
  tc = w / gr;
  tr = 8 * g + (w - tc * gr);
}

// SYMMETRIC MODE: the tiles (tr, tc) of an nr x nc rectangle of tiles with tc <= tr + band (band 0: on and below the diagonal - a Schur
// block, read back through (max, min) only; band 1: one tile above it as well - the pivot block, whose diagonal blocks of <= 64 pivots may
// straddle a tile boundary), numbered for the grouped order of nd_grouped: groups of 8 tile rows, column by column inside a group.
// Enumerating the allowed tiles - instead of launching the rectangle and returning from the others - keeps the eight XCDs' runs of
// consecutive tiles equally long (the first version skipped: the XCD that held the last rows of a 119 x 119 triangle had twice the
// average work and the halved flops showed as no gain at all near the root).  The loops are over <= nr / 8 groups: scalar work.
__host__ __device__ inline int nd_sym_group_count(int g, int nr, int nc, int band, int& rows_g, int& nfull) {
  rows_g = nr - 8 * g < 8 ? nr - 8 * g : 8;
  nfull = 8 * g + band + 1 < nc ? 8 * g + band + 1 : nc;  // columns every row of the group may use
  int cnt = nfull * rows_g;
  for (int c = nfull, len = rows_g - 1; c < nc && len > 0; ++c, --len) cnt += len;
  return cnt;
}
__host__ __device__ inline int nd_sym_tiles(int nr, int nc, int band) {
  int tot = 0, rows_g, nfull;
  for (int g = 0; 8 * g < nr; ++g) tot += nd_sym_group_count(g, nr, nc, band, rows_g, nfull);
  return tot;
}
__host__ __device__ inline void nd_sym_tile(int t, int nr, int nc, int band, int& tr, int& tc) {
  int g = 0, rows_g, nfull;
  for (;; ++g) {
    const int cnt = nd_sym_group_count(g, nr, nc, band, rows_g, nfull);
    if (t < cnt || 8 * (g + 1) >= nr) break;
    t -= cnt;
  }
  if (t < nfull * rows_g) {
    tc = t / rows_g;
    tr = 8 * g + (t - tc * rows_g);
    return;
  }
  t -= nfull * rows_g;
  int c = nfull, len = rows_g - 1;
  while (t >= len && len > 1) t -= len, ++c, --len;
  tc = c;
  tr = 8 * g + (rows_g - len) + t;  // rows of the group with r + band >= c: the last `len` ones
}

// The trailing update of an outer block in ONE launch (round 5): C is the L-shaped region rows / columns [o, M) of the front without
// its Schur block [Ps, M)^2 - region A = rows [o, Ps) x columns [o, M), region B = rows [Ps, M) x columns [o, Ps) - and blockIdx.y the
// running tile number over A then B (three launches of 400-900 tiles each left the epilogues of one launch uncovered by the MFMAs of
// the next).  nd_lshape_tiles gives the grid's y extent.
__host__ __device__ inline int nd_lshape_tiles(int TS, int o, int M, int Ps) {
  const int nc = (M - o + TS - 1) / TS, nrA = (Ps - o + TS - 1) / TS, nrB = (M - Ps + TS - 1) / TS;
  return nrA * nc + nrB * nrA;
}
__device__ __forceinline__ void nd_lshape_tile(int TS, int t, int o, int M, int Ps, int& r0, int& c0, int& rmax, int& cmax,
                                               int order = 0) {
  const int nc = (M - o + TS - 1) / TS, nrA = (Ps - o + TS - 1) / TS;
  int tr, tc;
  if (t < nrA * nc) {
    if (order)
      nd_grouped(t, nrA, nc, tr, tc);
    else
      tr = t / nc, tc = t % nc;
    r0 = o + TS * tr, c0 = o + TS * tc, rmax = Ps, cmax = M;
  } else {
    const int u = t - nrA * nc;
    if (order)
      nd_grouped(u, (M - Ps + TS - 1) / TS, nrA, tr, tc);
    else
      tr = u / nrA, tc = u % nrA;
    r0 = Ps + TS * tr, c0 = o + TS * tc, rmax = M, cmax = Ps;
  }
}

// C -= A B on the rectangle rows [r0g, r1g) x cols [c0g, c1g) of every front of the level, A = F[rows, k0:k1),
// B = F[k0:k1, cols); (32 WT) x (32 WT) tiles, 4 waves x (WT x WT) MFMA tiles of v_mfma_f64_16x16x4_f64, operands swapped
// (D^T = B^T A^T) so that the 16 lanes of an MFMA row write 128 contiguous bytes of C.  The next k-chunk is prefetched
// into registers while the current one feeds the matrix cores.
template <int WT, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_nd_gemm(double* __restrict__ arena, int64_t lev_off, int M, int r0g, int r1g,
                                                 int c0g, int c1g, int k0, int k1, int64_t store_off, int P, NdGatherCtx gc, int lsP, int order, int symskip) {
  constexpr int TS = 32 * WT;
  constexpr int NLD = ND_KC * TS / 256;  // elements of each operand a thread stages per chunk
  __shared__ double As[ND_KC][TS + 8];
  __shared__ double Bs[TS][ND_KC + 1];
  int r0, c0, rmax, cmax, front, t;
  nd_block_tile(order, (int)(gridDim.y * gridDim.z), front, t);
  if (symskip > 0) {  // the grid's y extent = nd_sym_tiles(...): only the allowed tiles exist
    int tr, tc;
    nd_sym_tile(order ? t : (int)blockIdx.y, (r1g - r0g + TS - 1) / TS, (c1g - c0g + TS - 1) / TS, symskip - 1, tr, tc);
    r0 = r0g + TS * tr, c0 = c0g + TS * tc, rmax = r1g, cmax = c1g;
  } else if (lsP < 0) {
    int tr = (int)blockIdx.y, tc = (int)blockIdx.z;
    if (order) nd_grouped(t, (int)gridDim.y, (int)gridDim.z, tr, tc);
    r0 = r0g + TS * tr, c0 = c0g + TS * tc, rmax = r1g, cmax = c1g;
  } else {
    nd_lshape_tile(TS, order ? t : (int)blockIdx.y, r0g, r1g, lsP, r0, c0, rmax, cmax, order);
  }
  if (r0 >= rmax || c0 >= cmax) return;
  // symskip < 0: the rectangle is launched whole and the tiles above the band return (fallback for grids beyond 65535 tiles)
  if (symskip < 0 && c0 > r0 + (-symskip - 1) * TS) return;
  double* F = arena + lev_off + (int64_t)front * M * M;  // C: working matrix
  const int64_t MP = (int64_t)M * P;
  const double* S = arena + store_off + (int64_t)front * (MP + (int64_t)P * (M - P));  // A, B: solved panels (compact store)
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const int wi = (wv >> 1) * 16 * WT, wj = (wv & 1) * 16 * WT;
  nd_v4d acc[WT][WT];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b) acc[a][b] = (nd_v4d){0.0, 0.0, 0.0, 0.0};
  double ra[NLD], rb[NLD];
  auto fetch = [&](int kc) {
    const int kn = min(ND_KC, k1 - kc);
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + 256 * q;
      const int i = idx % TS, k = idx / TS;
      ra[q] = (k < kn && r0 + i < rmax) ? S[(int64_t)(kc + k) * M + r0 + i] : 0.0;
      const int k2 = idx % ND_KC, j2 = idx / ND_KC, cj = c0 + j2;
      rb[q] = (k2 < kn && cj < cmax) ? S[cj < P ? (int64_t)cj * M + kc + k2 : MP + (int64_t)(cj - P) * P + kc + k2] : 0.0;
    }
  };
  fetch(k0);
  for (int kc = k0; kc < k1; kc += ND_KC) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + 256 * q;
      As[idx / TS][idx % TS] = ra[q];
      Bs[idx / ND_KC][idx % ND_KC] = rb[q];
    }
    __syncthreads();
    if (kc + ND_KC < k1) fetch(kc + ND_KC);
#pragma unroll
    for (int kk = 0; kk < ND_KC; kk += 4) {
      const int kq = kk + (l >> 4);
      double uf[WT], lf[WT];
#pragma unroll
      for (int t = 0; t < WT; ++t) {
        uf[t] = Bs[wj + 16 * t + (l & 15)][kq];
        lf[t] = As[kq][wi + 16 * t + (l & 15)];
      }
#pragma unroll
      for (int tj = 0; tj < WT; ++tj)
#pragma unroll
        for (int ti = 0; ti < WT; ++ti) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[tj], lf[ti], acc[tj][ti], 0, 0, 0);
    }
  }
  // D[m][n] = sum_k U[k][j=m] L[i=n][k]: lane l holds n = l&15 (row i of C), m = (l>>4) + 4*reg (column j of C)
  if (GATHER) {
    nd_gather_epilogue<WT, WT>(gc, arena, gc.f0 + front, F, M, r0 + wi, c0 + wj, rmax, cmax, l, acc);
    return;
  }
#pragma unroll
  for (int tj = 0; tj < WT; ++tj)
#pragma unroll
    for (int ti = 0; ti < WT; ++ti) {
      const int i = r0 + wi + 16 * ti + (l & 15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + wj + 16 * tj + (l >> 4) + 4 * reg;
        if (i < rmax && j < cmax) F[(int64_t)j * M + i] -= acc[tj][ti][reg];
      }
    }
}

// 128 x 128 tiles with EIGHT waves (wave tile 64 x 32 = 4 x 2 MFMA tiles): 64 accumulator VGPRs instead of 128, so that
// two workgroups = 4 waves per SIMD are resident (launch bounds: 4 waves per SIMD -> <= 128 VGPRs) and one wave's LDS staging and
// barriers hide behind three others' MFMAs; 1.5x the LDS reads per flop of the 4-wave version, still far from the LDS bound.
template <bool GATHER>
__global__ __launch_bounds__(512, 4) void k_nd_gemm8(double* __restrict__ arena, int64_t lev_off, int M, int r0g, int r1g,
                                                     int c0g, int c1g, int k0, int k1, int64_t store_off, int P, NdGatherCtx gc, int lsP, int order, int symskip) {
  constexpr int TS = 128, NT = 512;
  constexpr int NLD = ND_KC * TS / NT;  // 4 elements of each operand per thread and chunk
  __shared__ double As[ND_KC][TS + 8];
  __shared__ double Bs[TS][ND_KC + 1];
  int r0, c0, rmax, cmax, front, t;
  nd_block_tile(order, (int)(gridDim.y * gridDim.z), front, t);
  if (symskip > 0) {  // the grid's y extent = nd_sym_tiles(...): only the allowed tiles exist
    int tr, tc;
    nd_sym_tile(order ? t : (int)blockIdx.y, (r1g - r0g + TS - 1) / TS, (c1g - c0g + TS - 1) / TS, symskip - 1, tr, tc);
    r0 = r0g + TS * tr, c0 = c0g + TS * tc, rmax = r1g, cmax = c1g;
  } else if (lsP < 0) {
    int tr = (int)blockIdx.y, tc = (int)blockIdx.z;
    if (order) nd_grouped(t, (int)gridDim.y, (int)gridDim.z, tr, tc);
    r0 = r0g + TS * tr, c0 = c0g + TS * tc, rmax = r1g, cmax = c1g;
  } else {
    nd_lshape_tile(TS, order ? t : (int)blockIdx.y, r0g, r1g, lsP, r0, c0, rmax, cmax, order);
  }
  if (r0 >= rmax || c0 >= cmax) return;
  // symskip < 0: the rectangle is launched whole and the tiles above the band return (fallback for grids beyond 65535 tiles)
  if (symskip < 0 && c0 > r0 + (-symskip - 1) * TS) return;
  double* F = arena + lev_off + (int64_t)front * M * M;  // C: working matrix
  const int64_t MP = (int64_t)M * P;
  const double* S = arena + store_off + (int64_t)front * (MP + (int64_t)P * (M - P));  // A, B: solved panels
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const int wi = (wv >> 2) * 64, wj = (wv & 3) * 32;
  nd_v4d acc[2][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (nd_v4d){0.0, 0.0, 0.0, 0.0};
  double ra[NLD], rb[NLD];
  auto fetch = [&](int kc) {
    const int kn = min(ND_KC, k1 - kc);
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + NT * q;
      const int i = idx % TS, k = idx / TS;
      ra[q] = (k < kn && r0 + i < rmax) ? S[(int64_t)(kc + k) * M + r0 + i] : 0.0;
      const int k2 = idx % ND_KC, j2 = idx / ND_KC, cj = c0 + j2;
      rb[q] = (k2 < kn && cj < cmax) ? S[cj < P ? (int64_t)cj * M + kc + k2 : MP + (int64_t)(cj - P) * P + kc + k2] : 0.0;
    }
  };
  fetch(k0);
  for (int kc = k0; kc < k1; kc += ND_KC) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + NT * q;
      As[idx / TS][idx % TS] = ra[q];
      Bs[idx / ND_KC][idx % ND_KC] = rb[q];
    }
    __syncthreads();
    if (kc + ND_KC < k1) fetch(kc + ND_KC);
#pragma unroll
    for (int kk = 0; kk < ND_KC; kk += 4) {
      const int kq = kk + (l >> 4);
      double uf[2], lf[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) uf[t] = Bs[wj + 16 * t + (l & 15)][kq];
#pragma unroll
      for (int t = 0; t < 4; ++t) lf[t] = As[kq][wi + 16 * t + (l & 15)];
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[tj], lf[ti], acc[tj][ti], 0, 0, 0);
    }
  }
  if (GATHER) {
    nd_gather_epilogue<2, 4>(gc, arena, gc.f0 + front, F, M, r0 + wi, c0 + wj, rmax, cmax, l, acc);
    return;
  }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int i = r0 + wi + 16 * ti + (l & 15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + wj + 16 * tj + (l >> 4) + 4 * reg;
        if (i < rmax && j < cmax) F[(int64_t)j * M + i] -= acc[tj][ti][reg];
      }
    }
}


