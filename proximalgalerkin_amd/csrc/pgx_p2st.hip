// Structured operator apply for P2 on the uniform right-diagonal mesh (round 4) - gfx950.
//
// Reference: `obstacle_pg.py -p 2` assembles the P2 Newton matrix with DOLFINx and hands it to MUMPS
// (/root/reference/examples/01_obstacle_problem/obstacle_pg.py:68-70,125-139,288).  Here the Krylov solver and the residuals of the
// patch smoother apply J = [[aK, M], [M, -D(psi)]] to vectors; until round 3 always through the block-CSR kernel k_bspmv_bal
// (13 B per entry: column, one-byte (K, M) code, D; products staged in LDS; 0.45 of the HBM peak).
//
// On the structured mesh a P2 dof is one of four kinds - vertex (i, j), horizontal / vertical / diagonal edge of that vertex - and,
// away from the boundary, every row of a kind has the SAME columns relative to its own position (19 for a vertex, 9 for an edge)
// and the same K and M entries; only D(psi) varies.  So for the interior GROUPS (a vertex and its three edges):
//   * no column indices and no codes are read: a neighbour's dof index is the group's vertex index (or its first edge index) plus
//     a constant from a 46-entry table that the host derives from the CSR pattern of one interior group and checks on all of them;
//   * K and M are 46 pairs of kernel arguments;
//   * D(psi) is read from a structure-of-arrays copy Dst[entry][group] (k_p2st_pack, once per Newton step): consecutive lanes are
//     consecutive groups, every load is a full line;
//   * a thread owns a group: 4 rows, both fields, every load of a row issued before its first use - no LDS staging, no row sums.
// The frame (groups within two of the boundary - on a strip also of its cut lines -, where columns are missing or Dirichlet) keeps the CSR form: k_p2_rows_csr on the
// list of its rows.  Algorithmic bytes per group: 46 D values + 4 (u, psi) pairs read + 4 written = 496 B (CSR: 726 B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_internal.h"

namespace {

template <typename DT, int T, int NT>
__device__ __forceinline__ void p2st_row(const P2StTab& S, int off, size_t G, size_t g, int v, int eb, const DT* __restrict__ Dst,
                                         const double* __restrict__ xu, const double* __restrict__ xp, double& au, double& ap) {
  double d[NT], a[NT], b[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int e = off + k;
    const int idx = (S.isedge[e] ? eb : v) + S.delta[e];
    d[k] = (double)__builtin_nontemporal_load(Dst + (size_t)e * G + g);
    a[k] = xu[idx];
    b[k] = xp[idx];
  }
  double su = 0.0, sp = 0.0, tu = 0.0, tp = 0.0;  // two chains per field
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int e = off + k;
    if (k & 1) {
      tu += S.aK[e] * a[k] + S.M[e] * b[k];
      tp += S.M[e] * a[k] - d[k] * b[k];
    } else {
      su += S.aK[e] * a[k] + S.M[e] * b[k];
      sp += S.M[e] * a[k] - d[k] * b[k];
    }
  }
  au = su + tu;
  ap = sp + tp;
}

// y = J x (bu == nullptr) or y = b - J x on the interior groups i0 <= i < i0 + ni, j0 <= j < j0 + nj
template <typename DT>
__global__ void __launch_bounds__(256) k_p2st_apply(const P2StTab S, int nx, int nv, int i0, int ni, int j0, size_t G,
                                                    const DT* __restrict__ Dst, const double* __restrict__ xu,
                                                    const double* __restrict__ xp, const double* __restrict__ bu,
                                                    const double* __restrict__ bp, double* __restrict__ yu, double* __restrict__ yp) {
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= ni) return;
  const int i = i0 + li, j = j0 + blockIdx.y;
  const int v = j * (nx + 1) + i;
  const int eb = nv + j * (3 * nx + 1) + 3 * i;  // first edge dof of the group (interior rows: H, V, D at eb, eb + 1, eb + 2)
  const size_t g = (size_t)v;
  double au, ap;
  p2st_row<DT, 0, 19>(S, 0, G, g, v, eb, Dst, xu, xp, au, ap);
  if (bu) au = bu[v] - au, ap = bp[v] - ap;
  yu[v] = au, yp[v] = ap;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    p2st_row<DT, 1, 9>(S, 19 + 9 * t, G, g, v, eb, Dst, xu, xp, au, ap);
    const int r = eb + t;
    if (bu) au = bu[r] - au, ap = bp[r] - ap;
    yu[r] = au, yp[r] = ap;
  }
}

// The same with the iterate staged in LDS (the default): a block of BW groups of one row loads the (u, psi) pairs of the three vertex
// rows and three edge rows its stencils reach - coalesced, once - and every one of the 92 gathers of a group becomes an LDS read at
// base + lofs[entry] (the host has decoded every table entry into (dj, di, kind) with |dj|, |di| <= 1).
template <typename DT, int NT, int BW>
__device__ __forceinline__ void p2st_row_lds(const P2StTab& S, int off, size_t G, size_t g, const DT* __restrict__ Dst, const double2* xv,
                                             const double2* xe, double& au, double& ap) {
  double d[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) d[k] = (double)__builtin_nontemporal_load(Dst + (size_t)(off + k) * G + g);
  double su = 0.0, sp = 0.0, tu = 0.0, tp = 0.0;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int e = off + k;
    const double2 x = (S.isedge[e] ? xe : xv)[S.lofs[e]];
    if (k & 1) {
      tu += S.aK[e] * x.x + S.M[e] * x.y;
      tp += S.M[e] * x.x - d[k] * x.y;
    } else {
      su += S.aK[e] * x.x + S.M[e] * x.y;
      sp += S.M[e] * x.x - d[k] * x.y;
    }
  }
  au = su + tu;
  ap = sp + tp;
}

template <typename DT, int BW>
__global__ void __launch_bounds__(BW) k_p2st_apply_lds(const P2StTab S, int nx, int nv, int i0, int ni, int j0, size_t G,
                                                       const DT* __restrict__ Dst, const double* __restrict__ xu,
                                                       const double* __restrict__ xp, const double* __restrict__ bu,
                                                       const double* __restrict__ bp, double* __restrict__ yu, double* __restrict__ yp) {
  constexpr int VW = BW + 2, EW = 3 * VW;
  __shared__ double2 sv[3 * VW], se[3 * EW];
  const int tid = threadIdx.x;
  const int ib = i0 + blockIdx.x * BW, j = j0 + blockIdx.y;
  const int sx = nx + 1, erow = 3 * nx + 1;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int gj = j - 1 + r;
    for (int t = tid; t < VW; t += BW) {
      const int gi = ib - 1 + t;
      double2 x = make_double2(0.0, 0.0);
      if (gi <= nx) {
        const int v = gj * sx + gi;
        x = make_double2(xu[v], xp[v]);
      }
      sv[r * VW + t] = x;
    }
    for (int t = tid; t < EW; t += BW) {
      const int gi3 = 3 * (ib - 1) + t;  // position in the row's edge block
      double2 x = make_double2(0.0, 0.0);
      if (gi3 < erow) {
        const int e = nv + gj * erow + gi3;
        x = make_double2(xu[e], xp[e]);
      }
      se[r * EW + t] = x;
    }
  }
  __syncthreads();
  const int li = blockIdx.x * BW + tid;
  if (li >= ni) return;
  const int i = ib + tid;
  const int v = j * sx + i;
  const int eb = nv + j * erow + 3 * i;
  const size_t g = (size_t)v;
  const double2* xv = sv + tid;
  const double2* xe = se + 3 * tid;
  double au, ap;
  p2st_row_lds<DT, 19, BW>(S, 0, G, g, Dst, xv, xe, au, ap);
  if (bu) au = bu[v] - au, ap = bp[v] - ap;
  yu[v] = au, yp[v] = ap;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    p2st_row_lds<DT, 9, BW>(S, 19 + 9 * t, G, g, Dst, xv, xe, au, ap);
    const int r = eb + t;
    if (bu) au = bu[r] - au, ap = bp[r] - ap;
    yu[r] = au, yp[r] = ap;
  }
}

// CSR D(psi) -> Dst[entry][group] for the interior groups (DT = double or float)
template <typename DT>
__global__ void __launch_bounds__(256) k_p2st_pack(int nx, int nv, int i0, int ni, int j0, size_t G, const int32_t* __restrict__ rowptr,
                                                   const double* __restrict__ D, DT* __restrict__ Dst) {
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= ni) return;
  const int i = i0 + li, j = j0 + blockIdx.y;
  const int v = j * (nx + 1) + i;
  const int eb = nv + j * (3 * nx + 1) + 3 * i;
  const size_t g = (size_t)v;
  {
    const double* d = D + rowptr[v];
#pragma unroll
    for (int k = 0; k < 19; ++k) Dst[(size_t)k * G + g] = (DT)d[k];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const double* d = D + rowptr[eb + t];
#pragma unroll
    for (int k = 0; k < 9; ++k) Dst[(size_t)(19 + 9 * t + k) * G + g] = (DT)d[k];
  }
}

// do all interior groups carry the reference group's K and M entries?  fail[0] counts the entries that differ by more than tol
__global__ void __launch_bounds__(256) k_p2st_check(const P2StTab S, int nx, int nv, int i0, int ni, int j0, double alpha_ref, double tol,
                                                    const int32_t* __restrict__ rowptr, const double* __restrict__ K,
                                                    const double* __restrict__ M, int* __restrict__ fail) {
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= ni) return;
  const int i = i0 + li, j = j0 + blockIdx.y;
  const int v = j * (nx + 1) + i;
  const int eb = nv + j * (3 * nx + 1) + 3 * i;
  int bad = 0;
  for (int t = 0; t < 4; ++t) {
    const int row = t == 0 ? v : eb + t - 1, off = t == 0 ? 0 : 19 + 9 * (t - 1), nt = t == 0 ? 19 : 9;
    const int p = rowptr[row];
    for (int k = 0; k < nt; ++k) {
      if (fabs(alpha_ref * K[p + k] - S.aK[off + k]) > tol * S.kmax) ++bad;
      if (fabs(M[p + k] - S.M[off + k]) > tol * S.mmax) ++bad;
    }
  }
  if (bad) atomicAdd(fail, bad);
}

// the rows of the frame, CSR form, 16 lanes per row; semantics of k_bspmv_bal (Dirichlet columns of u flagged in the sign bit of
// colm, identity rows for Dirichlet u dofs)
__global__ void __launch_bounds__(256) k_p2_rows_csr(int nrows, const int32_t* __restrict__ rows, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colm, const double* __restrict__ K,
                                                     const double* __restrict__ M, const double* __restrict__ D, double alpha,
                                                     const uint8_t* __restrict__ mask, const double* __restrict__ xu,
                                                     const double* __restrict__ xp, const double* __restrict__ bu,
                                                     const double* __restrict__ bp, double* __restrict__ yu, double* __restrict__ yp) {
  const int gid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const int lane = threadIdx.x & 15;
  const bool live = gid < nrows;
  const int row = live ? rows[gid] : 0;
  double au = 0.0, ap = 0.0;
  if (live)
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 16) {
      const int c = colm[k], cc = c & 0x7fffffff;
      const double a = c >= 0 ? xu[cc] : 0.0, b = xp[cc];
      au += alpha * K[k] * a + M[k] * b;
      ap += M[k] * a - D[k] * b;
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    au += __shfl_xor(au, o, 16);
    ap += __shfl_xor(ap, o, 16);
  }
  if (!live || lane) return;
  if (mask[row]) au = xu[row];
  if (bu) au = bu[row] - au, ap = bp[row] - ap;
  yu[row] = au;
  yp[row] = ap;
}

}  // namespace

void pgxk_p2st_apply(hipStream_t st, const P2StTab& S, int nx, int nv, int i0, int ni, int j0, int nj, size_t G, const void* Dst, int f32,
                     const double* xu, const double* xp, const double* bu, const double* bp, double* yu, double* yp) {
  if (ni <= 0 || nj <= 0) return;
  if (S.lds) {
    constexpr int BW = 128;
    const dim3 grid((unsigned)((ni + BW - 1) / BW), (unsigned)nj);
    if (f32)
      hipLaunchKernelGGL((k_p2st_apply_lds<float, BW>), grid, dim3(BW), 0, st, S, nx, nv, i0, ni, j0, G, (const float*)Dst, xu, xp, bu, bp,
                         yu, yp);
    else
      hipLaunchKernelGGL((k_p2st_apply_lds<double, BW>), grid, dim3(BW), 0, st, S, nx, nv, i0, ni, j0, G, (const double*)Dst, xu, xp, bu, bp,
                         yu, yp);
    return;
  }
  const dim3 grid((unsigned)((ni + 255) / 256), (unsigned)nj);
  if (f32)
    hipLaunchKernelGGL(k_p2st_apply<float>, grid, dim3(256), 0, st, S, nx, nv, i0, ni, j0, G, (const float*)Dst, xu, xp, bu, bp, yu, yp);
  else
    hipLaunchKernelGGL(k_p2st_apply<double>, grid, dim3(256), 0, st, S, nx, nv, i0, ni, j0, G, (const double*)Dst, xu, xp, bu, bp, yu, yp);
}
void pgxk_p2st_pack(hipStream_t st, int nx, int nv, int i0, int ni, int j0, int nj, size_t G, const int32_t* rowptr, const double* D,
                    void* Dst, int f32) {
  if (ni <= 0 || nj <= 0) return;
  const dim3 grid((unsigned)((ni + 255) / 256), (unsigned)nj);
  if (f32)
    hipLaunchKernelGGL(k_p2st_pack<float>, grid, dim3(256), 0, st, nx, nv, i0, ni, j0, G, rowptr, D, (float*)Dst);
  else
    hipLaunchKernelGGL(k_p2st_pack<double>, grid, dim3(256), 0, st, nx, nv, i0, ni, j0, G, rowptr, D, (double*)Dst);
}
void pgxk_p2st_check(hipStream_t st, const P2StTab& S, int nx, int nv, int i0, int ni, int j0, int nj, double alpha_ref, double tol,
                     const int32_t* rowptr, const double* K, const double* M, int* fail) {
  if (ni <= 0 || nj <= 0) return;
  hipLaunchKernelGGL(k_p2st_check, dim3((unsigned)((ni + 255) / 256), (unsigned)nj), dim3(256), 0, st, S, nx, nv, i0, ni, j0, alpha_ref,
                     tol, rowptr, K, M, fail);
}
void pgxk_p2_rows_csr(hipStream_t st, int nrows, const int32_t* rows, const int32_t* rowptr, const int32_t* colm, const double* K,
                      const double* M, const double* D, double alpha, const uint8_t* mask, const double* xu, const double* xp,
                      const double* bu, const double* bp, double* yu, double* yp) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(k_p2_rows_csr, dim3((unsigned)(((int64_t)nrows * 16 + 255) / 256)), dim3(256), 0, st, nrows, rows, rowptr, colm, K, M,
                     D, alpha, mask, xu, xp, bu, bp, yu, yp);
}
