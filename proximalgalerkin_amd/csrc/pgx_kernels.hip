// HIP kernels for gfx950 (MI355X / CDNA4).  Everything here is fp64, HBM-bandwidth-bound work:
// no MFMA.  64-wide wavefronts are assumed throughout (shuffle widths, reductions).
//
// Kernel inventory (DESIGN.md section 5 holds the byte counts, timings and rooflines):
//   k_resid_fill_p1<WRITE_D>       fused, atomic-free residual + D(psi) fill, row-parallel through an LDS row image
//   k_fill_rows<MODE>              row-parallel fill of the K / M / D(psi) CSR blocks (setup, pgx_jacobian_fill)
//   k_bspmv_stream                 y = Jx for J=[[aK,M],[M,-D]] with ONE shared pattern (28 B/nnz), CSR-stream via LDS
//   k_bspmv<MODE,LPR>              lanes-per-row variant: fallback, and the smoother/residual on general meshes
//   k_st_smooth2<PRE|POST>         two collective-Jacobi sweeps per launch on LDS tiles (+ prolongation)
//   k_st_resid_restrict_t          b_c = P^T(b - Jx) in one launch;  k_st_apply<MODE> plain sweeps on mid levels
//   k_mg_tail_lds / k_mg_tail      whole V-cycle of all small levels in one launch
//   k_rap7 / k_rap7h, k_csr_to_stencil(_h), k_restrict, k_prolong_add   hierarchy set-up and transfers
//   k_multidot<NV>, k_axpy_dot, k_multiaxpy_scale<NV>, ...                3-pass CGS2 and Krylov vector kernels
//   k_observables (+ P2 twins in pgx_p2.hip)
#include "pgx_internal.h"
#include "pgx_stencil.h"
#include <algorithm>

#define WAVE 64

// hipFuncSetAttribute is per device: remember, per (kernel slot, device), whether the dynamic-LDS limit was raised
static bool first_use_on_device(int slot) {
  static bool done[4][64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (done[slot][dev]) return false;
  done[slot][dev] = true;
  return true;
}

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// non-temporal 16-B load (streamed-once data: Krylov basis slices)
__device__ __forceinline__ double2 ldnt2(const double2* p) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d v = __builtin_nontemporal_load((const v2d*)p);
  return make_double2(v.x, v.y);
}

// sum over the block; result valid in thread 0. `sm` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)(blockDim.x / WAVE); ++k) r += sm[k];
  return r;
}

// ------------------------------------------------------------------------------------------------
// P1 element geometry: J=[x1-x0, x2-x0], G = grad_ref(N) * J^-1
// ------------------------------------------------------------------------------------------------
struct P1Geom {
  double adet;     // |det J|
  double G[3][2];  // physical gradients of the three hat functions
};

__device__ __forceinline__ P1Geom p1_geom(double x0, double y0, double x1, double y1, double x2, double y2) {
  P1Geom g;
  const double J00 = x1 - x0, J01 = x2 - x0, J10 = y1 - y0, J11 = y2 - y0;
  const double det = J00 * J11 - J01 * J10;
  const double inv = 1.0 / det;
  g.adet = fabs(det);
  g.G[1][0] = J11 * inv;
  g.G[1][1] = -J01 * inv;
  g.G[2][0] = -J10 * inv;
  g.G[2][1] = J00 * inv;
  g.G[0][0] = -(g.G[1][0] + g.G[2][0]);
  g.G[0][1] = -(g.G[1][1] + g.G[2][1]);
  return g;
}

// ------------------------------------------------------------------------------------------------
// ORDER-2 GEOMETRY, degree-1 fields (round 5): P1 hat functions on the reference triangle, mapped by the quadratic cell map of a
// 6-node triangle - what the reference's default run computes on its own meshes (obstacle_pg.py -p 1 on the order-2 disk of
// generate_mesh_gmsh.py:30-33).  `geo` = [cell][quadrature point][5]: |det J|, then J^-1 row-major (d xi_k / d x_d), tabulated by
// the host (fem.Mesh.geometry_at) and shared with the P2 kernels (pgx_p2.hip).  Nothing is constant per cell any more: stiffness,
// mass, load and the latent terms are all sums over the form's quadrature rule, as FFCx generates them for a non-affine cell.
// Same row-parallel structure and summation order as the affine kernels above (no atomics, bitwise reproducible); these kernels
// only run on file meshes - the structured fast path has affine cells by construction.
// ------------------------------------------------------------------------------------------------
struct P1GeomQ {
  double wd;       // w_k |det J(x_k)|
  double G[3][2];  // physical gradients of the three hat functions at x_k
};
__device__ __forceinline__ P1GeomQ p1_geom_q(const double* __restrict__ geo, size_t cq, double w) {
  const double* p = geo + 5 * cq;
  P1GeomQ g;
  g.wd = w * p[0];
  // reference gradients (-1,-1), (1,0), (0,1) times J^-1
  g.G[1][0] = p[1];
  g.G[1][1] = p[2];
  g.G[2][0] = p[3];
  g.G[2][1] = p[4];
  g.G[0][0] = -(p[1] + p[3]);
  g.G[0][1] = -(p[2] + p[4]);
  return g;
}

__global__ void __launch_bounds__(PGX_BLOCK) k_bphi_c(int nc, const double* __restrict__ phi_q, QuadTab q,
                                                      double* __restrict__ stash, const double* __restrict__ geo) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  double b0 = 0, b1 = 0, b2 = 0;
  for (int k = 0; k < q.nq; ++k) {
    const double wp = q.w[k] * geo[5 * ((size_t)c * q.nq + k)] * phi_q[(size_t)c * q.nq + k];
    b0 += wp * q.N[k][0];
    b1 += wp * q.N[k][1];
    b2 += wp * q.N[k][2];
  }
  stash[4 * (size_t)c] = b0;
  stash[4 * (size_t)c + 1] = b1;
  stash[4 * (size_t)c + 2] = b2;
}

template <int MODE>
__global__ void __launch_bounds__(PGX_BLOCK) k_fill_rows_c(int n, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ v2c_ptr,
                                                           const int32_t* __restrict__ v2c_ent,
                                                           const int32_t* __restrict__ v2c_pos,
                                                           const int32_t* __restrict__ cells,
                                                           const double* __restrict__ psi, QuadTab q,
                                                           double* __restrict__ out, const double* __restrict__ geo) {
  extern __shared__ double acc[];
  const int i0 = blockIdx.x * PGX_BLOCK;
  const int i = i0 + threadIdx.x;
  const int iend = min(i0 + PGX_BLOCK, n);
  const int base = rowptr[i0];
  const int len = rowptr[iend] - base;
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) acc[k] = 0.0;
  __syncthreads();
  if (i < n) {
    double* row = acc + (rowptr[i] - base);
    const int ke = v2c_ptr[i + 1];
    for (int k = v2c_ptr[i]; k < ke; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 2, a = e & 3;
      const int pos = v2c_pos[k];
      double p0 = 0, p1 = 0, p2 = 0;
      if (MODE == 2) p0 = psi[cells[3 * c]], p1 = psi[cells[3 * c + 1]], p2 = psi[cells[3 * c + 2]];
      double d[3] = {0.0, 0.0, 0.0};
      for (int k2 = 0; k2 < q.nq; ++k2) {
        const P1GeomQ g = p1_geom_q(geo, (size_t)c * q.nq + k2, q.w[k2]);
        if (MODE == 0) {
#pragma unroll
          for (int b = 0; b < 3; ++b) d[b] += g.wd * (g.G[a][0] * g.G[b][0] + g.G[a][1] * g.G[b][1]);
        } else {
          double wa = g.wd * q.N[k2][a];
          if (MODE == 2) wa *= exp(p0 * q.N[k2][0] + p1 * q.N[k2][1] + p2 * q.N[k2][2]);
          d[0] += wa * q.N[k2][0];
          d[1] += wa * q.N[k2][1];
          d[2] += wa * q.N[k2][2];
        }
      }
      row[pos & 0xff] += d[0];
      row[(pos >> 8) & 0xff] += d[1];
      row[(pos >> 16) & 0xff] += d[2];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) out[base + k] = acc[k];
}

template <bool WRITE_D>
__global__ void __launch_bounds__(PGX_BLOCK) k_resid_fill_p1_c(int n, const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ v2c_ptr,
                                                               const int32_t* __restrict__ v2c_ent,
                                                               const int32_t* __restrict__ v2c_pos,
                                                               const int32_t* __restrict__ cells,
                                                               const uint8_t* __restrict__ mask,
                                                               const double* __restrict__ gbc,
                                                               const double* __restrict__ bphi,
                                                               const double* __restrict__ x,
                                                               const double* __restrict__ xk, double alpha, double f,
                                                               QuadTab q, double* __restrict__ F,
                                                               double* __restrict__ Dout, const double* __restrict__ geo) {
  extern __shared__ double acc[];
  const int i0 = blockIdx.x * PGX_BLOCK;
  const int i = i0 + threadIdx.x;
  const int iend = min(i0 + PGX_BLOCK, n);
  const int base = rowptr[i0];
  const int len = rowptr[iend] - base;
  if (WRITE_D) {
    for (int k = threadIdx.x; k < len; k += PGX_BLOCK) acc[k] = 0.0;
    __syncthreads();
  }
  if (i < n) {
    double* row = acc + (rowptr[i] - base);
    double Fu = 0.0, Fp = 0.0;
    const int ke = v2c_ptr[i + 1];
    for (int k = v2c_ptr[i]; k < ke; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 2, a = e & 3;
      const int v[3] = {cells[3 * c], cells[3 * c + 1], cells[3 * c + 2]};
      double u[3], p[3], dp[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        u[b] = mask[v[b]] ? gbc[v[b]] : x[v[b]];
        p[b] = x[n + v[b]];
        dp[b] = p[b] - xk[n + v[b]];
      }
      double d[3] = {0.0, 0.0, 0.0};
      for (int k2 = 0; k2 < q.nq; ++k2) {
        const P1GeomQ g = p1_geom_q(geo, (size_t)c * q.nq + k2, q.w[k2]);
        const double N0 = q.N[k2][0], N1 = q.N[k2][1], N2 = q.N[k2][2];
        const double uq = u[0] * N0 + u[1] * N1 + u[2] * N2;
        const double dq = dp[0] * N0 + dp[1] * N1 + dp[2] * N2;
        const double ex = exp(p[0] * N0 + p[1] * N1 + p[2] * N2);
        const double gx = u[0] * g.G[0][0] + u[1] * g.G[1][0] + u[2] * g.G[2][0];
        const double gy = u[0] * g.G[0][1] + u[1] * g.G[1][1] + u[2] * g.G[2][1];
        const double wa = g.wd * q.N[k2][a];
        Fu += g.wd * alpha * (g.G[a][0] * gx + g.G[a][1] * gy) + wa * (dq - alpha * f);
        Fp += wa * (uq - ex);
        d[0] += wa * ex * N0;
        d[1] += wa * ex * N1;
        d[2] += wa * ex * N2;
      }
      if (WRITE_D) {
        const int pos = v2c_pos[k];
        row[pos & 0xff] += d[0];
        row[(pos >> 8) & 0xff] += d[1];
        row[(pos >> 16) & 0xff] += d[2];
      }
    }
    F[i] = mask[i] ? x[i] - gbc[i] : Fu;
    F[n + i] = Fp - bphi[i];
  }
  if (WRITE_D) {
    __syncthreads();
    for (int k = threadIdx.x; k < len; k += PGX_BLOCK) Dout[base + k] = acc[k];
  }
}

__global__ void __launch_bounds__(PGX_BLOCK) k_observables_c(int nc, int n, const int32_t* __restrict__ cells,
                                                             const double* __restrict__ x, const double* __restrict__ xk,
                                                             double alpha, double f, QuadTab q, double* __restrict__ partials,
                                                             const double* __restrict__ geo) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    const int v[3] = {cells[3 * c], cells[3 * c + 1], cells[3 * c + 2]};
    double u[3], p[3], uk[3], pk[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[a] = x[v[a]];
      p[a] = x[n + v[a]];
      uk[a] = xk[v[a]];
      pk[a] = xk[n + v[a]];
    }
    for (int k = 0; k < q.nq; ++k) {
      const P1GeomQ g = p1_geom_q(geo, (size_t)c * q.nq + k, q.w[k]);
      const double wd = g.wd;
      const double gx = u[0] * g.G[0][0] + u[1] * g.G[1][0] + u[2] * g.G[2][0];
      const double gy = u[0] * g.G[0][1] + u[1] * g.G[1][1] + u[2] * g.G[2][1];
      const double hx = gx - (uk[0] * g.G[0][0] + uk[1] * g.G[1][0] + uk[2] * g.G[2][0]);
      const double hy = gy - (uk[0] * g.G[0][1] + uk[1] * g.G[1][1] + uk[2] * g.G[2][1]);
      const double uq = u[0] * q.N[k][0] + u[1] * q.N[k][1] + u[2] * q.N[k][2];
      const double pq = p[0] * q.N[k][0] + p[1] * q.N[k][1] + p[2] * q.N[k][2];
      const double ukq = uk[0] * q.N[k][0] + uk[1] * q.N[k][1] + uk[2] * q.N[k][2];
      const double pkq = pk[0] * q.N[k][0] + pk[1] * q.N[k][1] + pk[2] * q.N[k][2];
      s[0] += wd * (0.5 * (gx * gx + gy * gy) - f * uq);
      s[1] += wd * (pkq - pq) / alpha * uq;
      s[2] += wd * (uq < 0.0 ? -uq : 0.0);
      s[3] += wd * (pkq < pq ? (pq - pkq) / alpha : 0.0);
      const double du = uq - ukq;
      s[4] += wd * (hx * hx + hy * hy + du * du);
      const double de = exp(pq) - exp(pkq);
      s[5] += wd * de * de;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double r = block_sum(s[k], sm);
    if (threadIdx.x == 0) partials[blockIdx.x * 6 + k] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// b_phi = int phi w_i  (obstacle_pg.py:122 "- phi * w * dx"), assembled once at create
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PGX_BLOCK) k_bphi(int nc, const int32_t* __restrict__ cells,
                                                    const double* __restrict__ coords,
                                                    const double* __restrict__ phi_q, QuadTab q,
                                                    double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int v0 = cells[3 * c], v1 = cells[3 * c + 1], v2 = cells[3 * c + 2];
  const P1Geom g = p1_geom(coords[2 * v0], coords[2 * v0 + 1], coords[2 * v1], coords[2 * v1 + 1], coords[2 * v2],
                           coords[2 * v2 + 1]);
  double b0 = 0, b1 = 0, b2 = 0;
  for (int k = 0; k < q.nq; ++k) {
    const double wp = q.w[k] * phi_q[(size_t)c * q.nq + k];
    b0 += wp * q.N[k][0];
    b1 += wp * q.N[k][1];
    b2 += wp * q.N[k][2];
  }
  // parked at the index the vertex -> (cell, local vertex) lists use (cell * 4 + a); k_gather_ent sums per vertex in list order
  stash[4 * (size_t)c] = g.adet * b0;
  stash[4 * (size_t)c + 1] = g.adet * b1;
  stash[4 * (size_t)c + 2] = g.adet * b2;
}

// out[i] = sum of stash[ent[k]] over the dof's (cell, local index) list, in list order (ascending cells): the scatter-add of an
// assembly loop as a segmented reduction - no atomics, bitwise reproducible (pgx_scatter.h states the general case)
__global__ void __launch_bounds__(PGX_BLOCK) k_gather_ent(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent,
                                                          const double* __restrict__ stash, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += stash[ent[k]];
  out[i] = s;
}
void pgxk_gather_ent(hipStream_t st, int n, const int32_t* ptr, const int32_t* ent, const double* stash, double* out) {
  hipLaunchKernelGGL(k_gather_ent, dim3((n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, n, ptr, ent, stash, out);
}

void pgxk_bphi(hipStream_t st, int nc, int n, const int32_t* cells, const double* coords, const double* phi_q,
               QuadTab q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* bphi, const double* geo) {
  if (geo)
    hipLaunchKernelGGL(k_bphi_c, dim3((nc + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nc, phi_q, q, stash, geo);
  else
    hipLaunchKernelGGL(k_bphi, dim3((nc + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nc, cells, coords, phi_q,
                       q, stash);
  pgxk_gather_ent(st, n, v2c_ptr, v2c_ent, stash, bphi);
}

// ------------------------------------------------------------------------------------------------
// residual F(x) (obstacle_pg.py:116-124) with the BC contract of lvpp/problem.py:54-67.  P1 uses the fused
// row-parallel k_resid_fill_p1 below (the first version, a cell-parallel kernel with 6 fp64 atomics per cell,
// took 0.98 ms at 2048^2 and was not reproducible run to run).  k_residual_final is the last pass of the P2 path:
//   F_psi -= b_phi ;  F_u[bc] = u[bc] - g   (set_bc(F,bcs,x,-1))
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PGX_BLOCK) k_residual_final(int n, const uint8_t* __restrict__ mask,
                                                              const double* __restrict__ gbc,
                                                              const double* __restrict__ bphi,
                                                              const double* __restrict__ x, double* __restrict__ F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F[n + i] -= bphi[i];
  if (mask[i]) F[i] = x[i] - gbc[i];
}

void pgxk_residual_final(hipStream_t st, int n, const uint8_t* mask, const double* gbc, const double* bphi,
                         const double* x, double* F) {
  hipLaunchKernelGGL(k_residual_final, dim3((n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, n, mask, gbc,
                     bphi, x, F);
}

// ------------------------------------------------------------------------------------------------
// Row-parallel ("owner computes") fill of one scalar CSR block.  A block owns 256 consecutive rows,
// whose CSR value range [rowptr[i0], rowptr[i0+256]) is contiguous: contributions are summed in an
// LDS image of that range (each thread only touches its own row segment -> no atomics, bitwise
// reproducible), then the image is streamed out with fully coalesced stores.
//   MODE 0: K_ij = int grad phi_i . grad phi_j      MODE 1: M_ij      MODE 2: D_ij = int e^psi phi_i phi_j
// v2c_ent[k] = cell*4 + local index a of the row vertex in that cell
// v2c_pos[k] = positions (within the row's sorted column list) of the cell's 3 vertices, 8 bits each
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(PGX_BLOCK) k_fill_rows(int n, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ v2c_ptr,
                                                         const int32_t* __restrict__ v2c_ent,
                                                         const int32_t* __restrict__ v2c_pos,
                                                         const int32_t* __restrict__ cells,
                                                         const double* __restrict__ coords,
                                                         const double* __restrict__ psi, QuadTab q,
                                                         double* __restrict__ out) {
  extern __shared__ double acc[];
  const int i0 = blockIdx.x * PGX_BLOCK;
  const int i = i0 + threadIdx.x;
  const int iend = min(i0 + PGX_BLOCK, n);
  const int base = rowptr[i0];
  const int len = rowptr[iend] - base;
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) acc[k] = 0.0;
  __syncthreads();
  if (i < n) {
    double* row = acc + (rowptr[i] - base);
    const int ke = v2c_ptr[i + 1];
    for (int k = v2c_ptr[i]; k < ke; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 2, a = e & 3;
      const int pos = v2c_pos[k];
      const int v0 = cells[3 * c], v1 = cells[3 * c + 1], v2 = cells[3 * c + 2];
      const P1Geom g = p1_geom(coords[2 * v0], coords[2 * v0 + 1], coords[2 * v1], coords[2 * v1 + 1],
                               coords[2 * v2], coords[2 * v2 + 1]);
      double d[3];
      if (MODE == 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b) d[b] = 0.5 * g.adet * (g.G[a][0] * g.G[b][0] + g.G[a][1] * g.G[b][1]);
      } else if (MODE == 1) {
#pragma unroll
        for (int b = 0; b < 3; ++b) d[b] = g.adet * q.Mref[a][b];
      } else {
        const double p0 = psi[v0], p1 = psi[v1], p2 = psi[v2];
        d[0] = d[1] = d[2] = 0.0;
        for (int k2 = 0; k2 < q.nq; ++k2) {
          const double pq = p0 * q.N[k2][0] + p1 * q.N[k2][1] + p2 * q.N[k2][2];
          const double wa = q.w[k2] * exp(pq) * q.N[k2][a];
          d[0] += wa * q.N[k2][0];
          d[1] += wa * q.N[k2][1];
          d[2] += wa * q.N[k2][2];
        }
        d[0] *= g.adet;
        d[1] *= g.adet;
        d[2] *= g.adet;
      }
      row[pos & 0xff] += d[0];
      row[(pos >> 8) & 0xff] += d[1];
      row[(pos >> 16) & 0xff] += d[2];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) out[base + k] = acc[k];
}

// Fused residual + D(psi) fill, row-parallel (the Newton driver's kernel).  k_fill_rows<2> already evaluates
// exp(psi_h) at every quadrature point of every (vertex, incident cell) pair; the residual entries of that
// pair come out of the same loop at no extra exp cost (b_exp,a = row sum of the D_e row because the hat
// functions sum to 1).  One launch replaces memset + k_residual_p1 (6 fp64 atomics per cell, ~1 ms at
// 2048^2) + k_residual_final + k_fill_rows<2>, and the result is bitwise reproducible (no atomics).
template <bool WRITE_D, bool FRAME>
__global__ void __launch_bounds__(PGX_BLOCK) k_resid_fill_p1(int n, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ v2c_ptr,
                                                             const int32_t* __restrict__ v2c_ent,
                                                             const int32_t* __restrict__ v2c_pos,
                                                             const int32_t* __restrict__ cells,
                                                             const double* __restrict__ coords,
                                                             const uint8_t* __restrict__ mask,
                                                             const double* __restrict__ gbc,
                                                             const double* __restrict__ bphi,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ xk, double alpha, double f,
                                                             QuadTab q, int sx, int ny, double* __restrict__ F,
                                                             double* __restrict__ Dout) {
  // FRAME: only the boundary frame of an (sx x (ny+1))-vertex structured grid; the interior rows belong to k_resid_fill_grid.
  // Every row then zeroes and writes back its own segment of the LDS row image (no block-wide pass over rows it skipped).
  extern __shared__ double acc[];
  const int i0 = blockIdx.x * PGX_BLOCK;
  const int i = i0 + threadIdx.x;
  const int iend = min(i0 + PGX_BLOCK, n);
  const int base = rowptr[i0];
  const int len = rowptr[iend] - base;
  if (WRITE_D && !FRAME) {
    for (int k = threadIdx.x; k < len; k += PGX_BLOCK) acc[k] = 0.0;
    __syncthreads();
  }
  if (FRAME && i < n) {
    const int gi = i % sx, gj = i / sx;
    if (gi > 0 && gi < sx - 1 && gj > 0 && gj < ny) return;
  }
  if (i < n) {
    double* row = acc + (rowptr[i] - base);
    if (WRITE_D && FRAME)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) row[k - rowptr[i]] = 0.0;
    double Fu = 0.0, Fp = 0.0;
    const int ke = v2c_ptr[i + 1];
    for (int k = v2c_ptr[i]; k < ke; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 2, a = e & 3;
      const int v[3] = {cells[3 * c], cells[3 * c + 1], cells[3 * c + 2]};
      const P1Geom g = p1_geom(coords[2 * v[0]], coords[2 * v[0] + 1], coords[2 * v[1]], coords[2 * v[1] + 1],
                               coords[2 * v[2]], coords[2 * v[2] + 1]);
      double u[3], p[3], dp[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        u[b] = mask[v[b]] ? gbc[v[b]] : x[v[b]];
        p[b] = x[n + v[b]];
        dp[b] = p[b] - xk[n + v[b]];
      }
      double d[3] = {0.0, 0.0, 0.0};
      for (int k2 = 0; k2 < q.nq; ++k2) {
        const double pq = p[0] * q.N[k2][0] + p[1] * q.N[k2][1] + p[2] * q.N[k2][2];
        const double wa = q.w[k2] * exp(pq) * q.N[k2][a];
        d[0] += wa * q.N[k2][0];
        d[1] += wa * q.N[k2][1];
        d[2] += wa * q.N[k2][2];
      }
      const double gx = u[0] * g.G[0][0] + u[1] * g.G[1][0] + u[2] * g.G[2][0];
      const double gy = u[0] * g.G[0][1] + u[1] * g.G[1][1] + u[2] * g.G[2][1];
      const double Ku = 0.5 * g.adet * (g.G[a][0] * gx + g.G[a][1] * gy);
      const double Mdp = g.adet * (q.Mref[a][0] * dp[0] + q.Mref[a][1] * dp[1] + q.Mref[a][2] * dp[2]);
      const double Mu = g.adet * (q.Mref[a][0] * u[0] + q.Mref[a][1] * u[1] + q.Mref[a][2] * u[2]);
      Fu += alpha * Ku + Mdp - alpha * f * g.adet * q.mref[a];
      Fp += Mu - g.adet * (d[0] + d[1] + d[2]);
      if (WRITE_D) {
        const int pos = v2c_pos[k];
        row[pos & 0xff] += g.adet * d[0];
        row[(pos >> 8) & 0xff] += g.adet * d[1];
        row[(pos >> 16) & 0xff] += g.adet * d[2];
      }
    }
    F[i] = mask[i] ? x[i] - gbc[i] : Fu;
    F[n + i] = Fp - bphi[i];
    if (WRITE_D && FRAME)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) Dout[k] = row[k - rowptr[i]];
  }
  if (WRITE_D && !FRAME) {
    __syncthreads();
    for (int k = threadIdx.x; k < len; k += PGX_BLOCK) Dout[base + k] = acc[k];
  }
}

// ------------------------------------------------------------------------------------------------
// Structured-grid twin of k_resid_fill_p1 for the INTERIOR vertices of a uniform right-diagonal mesh: element-local
// evaluation with LDS-staged element blocks.  A workgroup owns 32 x 12 vertices.  Phase A: every triangle of the 33 x 13
// squares around them is evaluated ONCE - the six entries E_ab = sum_q w_q exp(psi_h(q)) N_a(q) N_b(q) of its symmetric
// latent block, 12 exp per triangle instead of the 36 the row-parallel kernel spends (each of a triangle's three rows
// re-evaluates it) - and parked in LDS.  Phase B: a thread per vertex gathers its six triangles in a fixed order (no
// atomics, bitwise reproducible): the seven entries of the D(psi) row, written in CSR order (SW, S, W, C, E, N, NE - the
// sorted columns of an interior row), their sum (= int exp(psi_h) phi_i), and the K / M parts of both residual rows from the
// seven constants of the uniform stencils.  Triangles of square (a, b): LR {(a,b), (a+1,b), (a+1,b+1)} and
// UL {(a,b), (a+1,b+1), (a,b+1)}; any symmetric quadrature rule gives the same E for any labelling of a triangle's vertices.
// The boundary frame (first / last row and column) keeps the general kernel (k_resid_fill_p1<., true>).
// ------------------------------------------------------------------------------------------------
#define PGX_RF_TX 32
#define PGX_RF_TY 12
template <bool WRITE_D>
__global__ void __launch_bounds__(PGX_RF_TX * PGX_RF_TY) k_resid_fill_grid(
    int nx, int ny, int n, const int32_t* __restrict__ rowptr, const double* __restrict__ coords, const uint8_t* __restrict__ mask,
    const double* __restrict__ gbc, const double* __restrict__ bphi, const double* __restrict__ x, const double* __restrict__ xk,
    double alpha, double f, QuadTab q, StConst sc, double* __restrict__ F, double* __restrict__ Dout,
    dsten_t* __restrict__ Sh) {
  constexpr int TX = PGX_RF_TX, TY = PGX_RF_TY, IW = TX + 2, IH = TY + 2, SW = TX + 1, SH = TY + 1;
  __shared__ double iu[IH * IW], ip[IH * IW], idp[IH * IW];
  __shared__ double E[2 * SW * SH][6];  // [2 * square + triangle][00, 11, 22, 01, 02, 12]
  const int sx = nx + 1;
  const int ntx = (nx - 1 + TX - 1) / TX;
  const int i1 = 1 + ((int)blockIdx.x % ntx) * TX, j1 = 1 + ((int)blockIdx.x / ntx) * TY;  // first vertex of the tile
  const int tid = threadIdx.x;
  for (int t = tid; t < IH * IW; t += TX * TY) {  // images start one vertex before the tile
    const int gi = i1 - 1 + t % IW, gj = j1 - 1 + t / IW;
    double u = 0.0, p = 0.0, dp = 0.0;
    if (gi <= nx && gj <= ny) {
      const int v = gj * sx + gi;
      u = mask[v] ? gbc[v] : x[v];
      p = x[n + v];
      dp = p - xk[n + v];
    }
    iu[t] = u;
    ip[t] = p;
    idp[t] = dp;
  }
  __syncthreads();
  for (int t = tid; t < 2 * SW * SH; t += TX * TY) {
    const int sq = t >> 1, a = sq % SW, b = sq / SW;
    double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i1 - 1 + a < nx && j1 - 1 + b < ny) {  // the square exists
      const int o = b * IW + a;
      const double p0 = ip[o], p1 = (t & 1) ? ip[o + IW + 1] : ip[o + 1], p2 = (t & 1) ? ip[o + IW] : ip[o + IW + 1];
      for (int k = 0; k < q.nq; ++k) {
        const double n0 = q.N[k][0], n1 = q.N[k][1], n2 = q.N[k][2];
        const double w = q.w[k] * exp(p0 * n0 + p1 * n1 + p2 * n2);
        e[0] += w * n0 * n0;
        e[1] += w * n1 * n1;
        e[2] += w * n2 * n2;
        e[3] += w * n0 * n1;
        e[4] += w * n0 * n2;
        e[5] += w * n1 * n2;
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) E[t][c] = e[c];
  }
  __syncthreads();
  const int li = tid % TX, lj = tid / TX, gi = i1 + li, gj = j1 + lj;
  if (gi >= nx || gj >= ny) return;  // interior vertices only: 1 <= gi <= nx - 1, 1 <= gj <= ny - 1
  const int v = gj * sx + gi;
  const double adet = fabs((coords[2] - coords[0]) * (coords[2 * sx + 1] - coords[1]));  // uniform mesh: |det J| of every triangle
  // squares around the vertex in tile-square coordinates: SW = (li, lj), SE = (li+1, lj), NW = (li, lj+1), NE = (li+1, lj+1)
  const double* swl = E[2 * (lj * SW + li)];
  const double* swu = E[2 * (lj * SW + li) + 1];
  const double* seu = E[2 * (lj * SW + li + 1) + 1];
  const double* nwl = E[2 * ((lj + 1) * SW + li)];
  const double* nel = E[2 * ((lj + 1) * SW + li + 1)];
  const double* neu = E[2 * ((lj + 1) * SW + li + 1) + 1];
  // E index: 0:00 1:11 2:22 3:01 4:02 5:12.  Local index of the vertex: SW-LR 2, SW-UL 1, SE-UL 2, NW-LR 1, NE-LR 0, NE-UL 0
  const double dC = adet * (((swl[2] + swu[1]) + (seu[2] + nwl[1])) + (nel[0] + neu[0]));
  const double dSW = adet * (swl[4] + swu[3]);  // SW-LR (2,0), SW-UL (1,0)
  const double dS = adet * (swl[5] + seu[4]);   // SW-LR (2,1), SE-UL (2,0)
  const double dW = adet * (swu[5] + nwl[3]);   // SW-UL (1,2), NW-LR (1,0)
  const double dE = adet * (seu[5] + nel[3]);   // SE-UL (2,1), NE-LR (0,1)
  const double dN = adet * (nwl[5] + neu[4]);   // NW-LR (1,2), NE-UL (0,2)
  const double dNE = adet * (nel[4] + neu[3]);  // NE-LR (0,2), NE-UL (0,1)
  if (WRITE_D) {
    if (Dout) {  // nullptr: the Newton loop of a matrix-free handle - nobody reads the CSR form of the interior rows (56 B per vertex)
      double* d = Dout + rowptr[v];
      d[0] = dSW;
      d[1] = dS;
      d[2] = dW;
      d[3] = dC;
      d[4] = dE;
      d[5] = dN;
      d[6] = dNE;
    }
    if (Sh) {  // the multigrid's half-stored stencil of the same row: centre + the three forward links
      Sh[v] = (dsten_t)dC;
      Sh[(size_t)n + v] = (dsten_t)dE;
      Sh[2 * (size_t)n + v] = (dsten_t)dN;
      Sh[3 * (size_t)n + v] = (dsten_t)dNE;
    }
  }
  const int o = (lj + 1) * IW + li + 1;
  const int off[7] = {0, 1, -1, IW, -IW, IW + 1, -IW - 1};
  double Ku = 0.0, Mdp = 0.0, Mu = 0.0;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    Ku += sc.K[t] * iu[o + off[t]];
    Mdp += sc.M[t] * idp[o + off[t]];
    Mu += sc.M[t] * iu[o + off[t]];
  }
  const double load = adet * 2.0 * (q.mref[0] + q.mref[1] + q.mref[2]);  // int phi_i over the six triangles
  F[v] = mask[v] ? x[v] - gbc[v] : alpha * Ku + Mdp - alpha * f * load;
  F[n + v] = Mu - (((dSW + dS) + (dW + dC)) + ((dE + dN) + dNE)) - bphi[v];
}

void pgxk_resid_fill_p1(hipStream_t st, int write_d, int n, size_t lds_bytes, const int32_t* rowptr,
                        const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells,
                        const double* coords, const uint8_t* mask, const double* gbc, const double* bphi,
                        const double* x, const double* xk, double alpha, double f, QuadTab q, double* F, double* Dout,
                        const double* geo) {
  dim3 grid((n + PGX_BLOCK - 1) / PGX_BLOCK), block(PGX_BLOCK);
  if (geo) {
    if (write_d)
      hipLaunchKernelGGL((k_resid_fill_p1_c<true>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, mask, gbc,
                         bphi, x, xk, alpha, f, q, F, Dout, geo);
    else
      hipLaunchKernelGGL((k_resid_fill_p1_c<false>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, mask, gbc,
                         bphi, x, xk, alpha, f, q, F, Dout, geo);
    return;
  }
  if (write_d)
    hipLaunchKernelGGL((k_resid_fill_p1<true, false>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells,
                       coords, mask, gbc, bphi, x, xk, alpha, f, q, 0, 0, F, Dout);
  else
    hipLaunchKernelGGL((k_resid_fill_p1<false, false>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells,
                       coords, mask, gbc, bphi, x, xk, alpha, f, q, 0, 0, F, Dout);
}

// structured uniform mesh (GridLevel L = the finest level): boundary frame through the general kernel, interior through
// k_resid_fill_grid
void pgxk_resid_fill_grid(hipStream_t st, int write_d, const GridLevel& L, size_t lds_bytes, const int32_t* rowptr,
                          const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells,
                          const double* coords, const uint8_t* mask, const double* gbc, const double* bphi, const double* x,
                          const double* xk, double alpha, double f, QuadTab q, double* F, double* Dout, int write_sh) {
  const int n = L.n, sx = L.nx + 1;
  dim3 grid((n + PGX_BLOCK - 1) / PGX_BLOCK), block(PGX_BLOCK);
  if (write_d)
    hipLaunchKernelGGL((k_resid_fill_p1<true, true>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells,
                       coords, mask, gbc, bphi, x, xk, alpha, f, q, sx, L.ny, F, Dout);
  else
    hipLaunchKernelGGL((k_resid_fill_p1<false, true>), grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells,
                       coords, mask, gbc, bphi, x, xk, alpha, f, q, sx, L.ny, F, Dout);
  if (L.nx < 2 || L.ny < 2) return;
  const int ntx = (L.nx - 1 + PGX_RF_TX - 1) / PGX_RF_TX, nty = (L.ny - 1 + PGX_RF_TY - 1) / PGX_RF_TY;
  const StConst sc = make_stconst(L);
  if (write_d)
    hipLaunchKernelGGL(k_resid_fill_grid<true>, dim3(ntx * nty), dim3(PGX_RF_TX * PGX_RF_TY), 0, st, L.nx, L.ny, n, rowptr, coords,
                       mask, gbc, bphi, x, xk, alpha, f, q, sc, F, write_sh == 2 ? nullptr : Dout, write_sh ? L.Dh : nullptr);
  else
    hipLaunchKernelGGL(k_resid_fill_grid<false>, dim3(ntx * nty), dim3(PGX_RF_TX * PGX_RF_TY), 0, st, L.nx, L.ny, n, rowptr, coords,
                       mask, gbc, bphi, x, xk, alpha, f, q, sc, F, Dout, nullptr);
}

void pgxk_fill_rows(hipStream_t st, int mode, int n, size_t lds_bytes, const int32_t* rowptr, const int32_t* v2c_ptr,
                    const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells, const double* coords,
                    const double* psi, QuadTab q, double* out, const double* geo) {
  dim3 grid((n + PGX_BLOCK - 1) / PGX_BLOCK), block(PGX_BLOCK);
  if (geo) {
    if (mode == 0)
      hipLaunchKernelGGL(k_fill_rows_c<0>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, psi, q, out, geo);
    else if (mode == 1)
      hipLaunchKernelGGL(k_fill_rows_c<1>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, psi, q, out, geo);
    else
      hipLaunchKernelGGL(k_fill_rows_c<2>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, psi, q, out, geo);
    return;
  }
  if (mode == 0)
    hipLaunchKernelGGL(k_fill_rows<0>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, coords,
                       psi, q, out);
  else if (mode == 1)
    hipLaunchKernelGGL(k_fill_rows<1>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, coords,
                       psi, q, out);
  else
    hipLaunchKernelGGL(k_fill_rows<2>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cells, coords,
                       psi, q, out);
}

// ------------------------------------------------------------------------------------------------
// Block-CSR SpMV for the Newton matrix  J = [[aK, M],[M, -D]]  with ONE shared scalar pattern.
// colm[k] = column | (column is a Dirichlet dof of u) << 31.  Dirichlet rows/cols of the u block act
// as identity (lvpp/problem.py:69-77 + dolfinx assemble_matrix(bcs) semantics):
//   y_u[i]   = bc_i ? x_u[i] : sum_j aK_ij x~_u[j] + M_ij x_psi[j]        x~_u = x_u with bc entries zeroed
//   y_psi[i] =                 sum_j  M_ij x~_u[j] - D_ij x_psi[j]
// LPR lanes cooperate on a row (P1: ~7 nnz/row); consecutive lanes read consecutive CSR entries, so
// the three value streams + the column stream are read coalesced.
// MODE 0: y = Jx   MODE 1: y = b - Jx   MODE 2: y = x + omega * Binv (b - Jx)   (collective Jacobi:
//   Binv = inverse of the vertex 2x2 block [[a,b],[b,-d]], det = -ad-b^2 < 0; bc rows use omega=1)
// ------------------------------------------------------------------------------------------------
template <int MODE, int LPR>
__global__ void __launch_bounds__(PGX_BLOCK) k_bspmv(int n, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colm,
                                                     const double* __restrict__ K, const double* __restrict__ M,
                                                     const double* __restrict__ D, double alpha,
                                                     const double* __restrict__ xu, const double* __restrict__ xp,
                                                     const double* __restrict__ bu, const double* __restrict__ bp,
                                                     double omega, int first, double* __restrict__ yu,
                                                     double* __restrict__ yp) {
  const int t = xcd_block(blockIdx.x, gridDim.x, first >> 1) * PGX_BLOCK + threadIdx.x;
  const int row = t / LPR, lane = t % LPR;
  const bool live = row < n;
  int s = 0, e = 0;
  if (live) {
    s = rowptr[row];
    e = rowptr[row + 1];
  }
  double au = 0.0, ap = 0.0, da = 0.0, dm = 0.0, dd = 0.0;
  int rowbc = 0;
  const bool skip = (MODE == 2) && (first & 1);
  for (int k = s + lane; k < e; k += LPR) {
    const int cm = colm[k];
    const int c = cm & 0x7fffffff;
    const double kv = K[k], mv = M[k], dv = D[k];
    if (c == row) {
      rowbc = cm < 0;
      da = alpha * kv;
      dm = mv;
      dd = dv;
    }
    if (!skip) {
      const double xuv = (cm < 0) ? 0.0 : xu[c];
      const double xpv = xp[c];
      au += alpha * kv * xuv + mv * xpv;
      ap += mv * xuv - dv * xpv;
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) {
    au += __shfl_xor(au, o, LPR);
    ap += __shfl_xor(ap, o, LPR);
    if (MODE == 2) {
      da += __shfl_xor(da, o, LPR);
      dm += __shfl_xor(dm, o, LPR);
      dd += __shfl_xor(dd, o, LPR);
    }
    rowbc |= __shfl_xor(rowbc, o, LPR);
  }
  if (!live || lane != 0) return;
  const double xur = skip ? 0.0 : xu[row];
  if (rowbc) au = xur;
  if (MODE == 0) {
    yu[row] = au;
    yp[row] = ap;
  } else if (MODE == 1) {
    yu[row] = bu[row] - au;
    yp[row] = bp[row] - ap;
  } else {
    const double su = bu[row] - au, sp = bp[row] - ap;
    const double xpr = skip ? 0.0 : xp[row];
    double a = da, b = dm;
    double om_u = omega;
    if (rowbc) {
      a = 1.0;
      b = 0.0;
      om_u = 1.0;
    }
    const double det = -a * dd - b * b;
    double du = 0.0, dpsi = 0.0;
    if (det != 0.0) {
      du = (-dd * su - b * sp) / det;
      dpsi = (-b * su + a * sp) / det;
    } else if (rowbc) {
      du = su;
    }
    yu[row] = xur + om_u * du;
    yp[row] = xpr + omega * dpsi;
  }
}

// "CSR-stream" form of y = Jx: a block owns 256 consecutive rows, i.e. ONE contiguous range of the CSR
// arrays.  Phase 1: all lanes stream that range (column + 3 value streams, unit stride, no idle lanes, no
// per-row loop), gather x, and park the two products per nonzero in LDS.  Phase 2: one thread per row sums
// its LDS segment.  Compared with k_bspmv<0,8>: no 1-in-8 idle lane, no shuffles, 7 independent 4-stream
// loads in flight per lane, and 32x fewer waves to launch.
__global__ void __launch_bounds__(PGX_BLOCK) k_bspmv_stream(int n, int cap, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colm,
                                                            const double* __restrict__ K,
                                                            const double* __restrict__ M,
                                                            const double* __restrict__ D, double alpha,
                                                            const uint8_t* __restrict__ mask,
                                                            const double* __restrict__ xu,
                                                            const double* __restrict__ xp, int remap,
                                                            double* __restrict__ yu, double* __restrict__ yp) {
  extern __shared__ double sprod[];  // [2][cap]
  __shared__ int srp[PGX_BLOCK + 1];
  const int r0 = xcd_block(blockIdx.x, gridDim.x, remap) * PGX_BLOCK;
  const int tid = threadIdx.x;
  srp[tid] = rowptr[min(r0 + tid, n)];
  if (tid == 0) srp[PGX_BLOCK] = rowptr[min(r0 + PGX_BLOCK, n)];
  __syncthreads();
  const int base = srp[0];
  const int len = srp[PGX_BLOCK] - base;
  const int32_t* cb = colm + base;
  const double *Kb = K + base, *Mb = M + base, *Db = D + base;
  double* su = sprod;
  double* sp = sprod + cap;
#pragma unroll 4
  for (int k = tid; k < len; k += PGX_BLOCK) {
    // the CSR streams are read exactly once: non-temporal loads keep them from evicting the gathered x from L2
    const int cm = __builtin_nontemporal_load(cb + k);
    const int c = cm & 0x7fffffff;
    const double kv = __builtin_nontemporal_load(Kb + k), mv = __builtin_nontemporal_load(Mb + k),
                 dv = __builtin_nontemporal_load(Db + k);
    const double xuv = (cm < 0) ? 0.0 : xu[c];
    const double xpv = xp[c];
    su[k] = alpha * kv * xuv + mv * xpv;
    sp[k] = mv * xuv - dv * xpv;
  }
  __syncthreads();
  const int row = r0 + tid;
  if (row >= n) return;
  double au = 0.0, ap = 0.0;
  for (int k = srp[tid] - base, e = srp[tid + 1] - base; k < e; ++k) {
    au += su[k];
    ap += sp[k];
  }
  if (mask[row]) au = xu[row];
  __builtin_nontemporal_store(au, yu + row);
  __builtin_nontemporal_store(ap, yp + row);
}

// k_bspmv_bal (round 3): the CSR-stream kernel with NNZ-balanced blocks.  k_bspmv_stream gives every block 256 rows and the LDS of
// the densest block: on P2 (19 nnz per vertex row, 9 per edge row) that is 78 KB per block = 2 blocks = 8 waves per CU, and the
// kernel sat at 0.33 of the HBM peak.  Here the host cuts the rows into blocks of at most CAP entries
// (blk[b] .. blk[b+1], never more than 256 rows), so every block parks <= CAP = 1536 products per field: 24 KB of LDS, 6 blocks per CU,
// equal work per block.  bu != nullptr: the residual b - J x instead of J x (the patch smoother's sweeps, pgx_patch.hip).
//
// DICT (round 3): on a uniform mesh the constant matrices K and M hold a few dozen distinct (K_ij, M_ij) pairs (one per kind of row
// and link, up to rounding).  The host finds them (pgxk_dict_assign; entries are rounded to a grid of 2^-40 of the largest entry -
// finer than the 1e-11 tolerance of the P1 levels' uniform stencils) and the kernel reads ONE BYTE per entry - an index into a 256-entry table in LDS - instead of two
// doubles: 13 B per entry (column, code, D) instead of 28.
// DF (round 4): D(psi) read from a FLOAT copy (9 instead of 13 B per entry) - for the residuals INSIDE the P2 two-level cycle, a
// preconditioner within FGMRES; the Krylov solver's own operator apply and every true residual read the fp64 array.
template <bool DICT, bool DF>
__global__ void __launch_bounds__(PGX_BLOCK) k_bspmv_bal(int n, const int32_t* __restrict__ blk, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colm, const double* __restrict__ K,
                                                         const double* __restrict__ M, const uint8_t* __restrict__ code,
                                                         const double2* __restrict__ table, const double* __restrict__ D,
                                                         const float* __restrict__ Df, double alpha,
                                                         const uint8_t* __restrict__ mask, const double* __restrict__ xu,
                                                         const double* __restrict__ xp, const double* __restrict__ bu,
                                                         const double* __restrict__ bp, int remap, double* __restrict__ yu,
                                                         double* __restrict__ yp) {
  __shared__ double su[PGX_BAL_CAP], sp[PGX_BAL_CAP];
  __shared__ int srp[PGX_BLOCK + 1];  // row ENDS relative to the block's first entry (srp[0] = 0)
  __shared__ double2 stab[DICT ? 256 : 1];
  const int b = xcd_block(blockIdx.x, gridDim.x, remap);
  // blk[2 b] = first row, blk[2 b + 1] = its rowptr: ONE (scalar) load tells the block its rows and its entry range, so that the row
  // pointers, the CSR streams and the per-row data of the last phase are all requested together (the version that read
  // blk -> rowptr -> streams -> gathers -> [barrier] -> b, mask in sequence: 0.90 ms per apply at 2048^2 P2, this one 0.87 ms; the
  // PMC profile - waves waiting 85 % of their 7.8 us life, about two memory instructions in flight per CU - points at the two
  // 8-byte gathers of x per entry, 20-40 cache lines per wave instruction, as what the kernel waits for: DESIGN.md section 5).
  const int r0 = blk[2 * b], base = blk[2 * b + 1], nr = blk[2 * b + 2] - r0, len = blk[2 * b + 3] - base;
  const int tid = threadIdx.x;
  double2 tv = make_double2(0.0, 0.0);
  if (DICT) tv = table[tid];  // first in the load queue: its wait does not cover the streams below
  const int row = r0 + tid;
  const bool live = tid < nr;
  int rend = 0;
  uint8_t mk = 0;
  double xur = 0.0, bur = 0.0, bpr = 0.0;
  if (live) {
    rend = rowptr[row + 1] - base;
    mk = mask[row];
    xur = xu[row];
    if (bu) {
      bur = bu[row];
      bpr = bp[row];
    }
  }
  const int32_t* cb = colm + base;
  const double *Kb = K + base, *Mb = M + base, *Db = D + base;
  const float* Dfb = Df + base;
  const uint8_t* qb = code + base;
  // A block holds at most PGX_BAL_CAP entries = IT per thread: all IT column / code / value loads are issued before the first
  // gather of x and all gathers before the first product
  constexpr int IT = PGX_BAL_CAP / PGX_BLOCK;
  static_assert(IT * PGX_BLOCK == PGX_BAL_CAP, "block capacity must be a multiple of the block size");
  static_assert(PGX_BLOCK >= 256, "one table entry per thread");
  int cm[IT];
  double kv[IT], mv[IT], dv[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int k = tid + i * PGX_BLOCK;
    cm[i] = 0;
    kv[i] = mv[i] = dv[i] = 0.0;
    if (k < len) {
      cm[i] = __builtin_nontemporal_load(cb + k);
      dv[i] = DF ? (double)__builtin_nontemporal_load(Dfb + k) : __builtin_nontemporal_load(Db + k);
      if (DICT) {
        kv[i] = (double)__builtin_nontemporal_load(qb + k);  // the code, parked in kv until the table is read below
      } else {
        kv[i] = __builtin_nontemporal_load(Kb + k);
        mv[i] = __builtin_nontemporal_load(Mb + k);
      }
    }
  }
  if (DICT) {
    if (tid < 256) stab[tid] = tv;
    __syncthreads();  // early: the streams are still in flight
  }
  double xuv[IT], xpv[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int k = tid + i * PGX_BLOCK;
    const int c = cm[i] & 0x7fffffff;
    xuv[i] = (k < len && cm[i] >= 0) ? xu[c] : 0.0;
    xpv[i] = (k < len) ? xp[c] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int k = tid + i * PGX_BLOCK;
    if (k < len) {
      double kk = kv[i], mm = mv[i];
      if (DICT) {
        const double2 km = stab[(int)kv[i]];
        kk = km.x;
        mm = km.y;
      }
      su[k] = alpha * kk * xuv[i] + mm * xpv[i];
      sp[k] = mm * xuv[i] - dv[i] * xpv[i];
    }
  }
  if (tid == 0) srp[0] = 0;
  if (live) srp[tid + 1] = rend;
  __syncthreads();
  if (!live) return;
  double au = 0.0, ap = 0.0;
  for (int k = srp[tid], e = rend; k < e; ++k) {
    au += su[k];
    ap += sp[k];
  }
  if (mk) au = xur;
  if (bu) {
    au = bur - au;
    ap = bpr - ap;
  }
  __builtin_nontemporal_store(au, yu + row);
  __builtin_nontemporal_store(ap, yp + row);
}

void pgxk_bspmv_bal(hipStream_t st, int n, int nblk, const int32_t* blk, const int32_t* rowptr, const int32_t* colm,
                    const double* K, const double* M, const uint8_t* code, const double* table, const double* D, double alpha,
                    const uint8_t* mask, const double* xu, const double* xp, const double* bu, const double* bp, int remap,
                    double* yu, double* yp, const float* Df) {
#define PGX_BAL(DI, FL)                                                                                                       \
  hipLaunchKernelGGL((k_bspmv_bal<DI, FL>), dim3(nblk), dim3(PGX_BLOCK), 0, st, n, blk, rowptr, colm, K, M, code,            \
                     (const double2*)table, D, Df, alpha, mask, xu, xp, bu, bp, remap, yu, yp)
  if (code && table) {
    if (Df) PGX_BAL(true, true); else PGX_BAL(true, false);
  } else {
    if (Df) PGX_BAL(false, true); else PGX_BAL(false, false);
  }
#undef PGX_BAL
}

// code[k] = index of the table entry that equals (K[k], M[k]) rounded to the grid (tk, tm); entries without one are counted in
// fail[0] and the rounded values of the first `cap` of them listed in fail_v (the host adds them to the table and calls again)
__global__ void __launch_bounds__(256) k_dict_assign(int64_t nnz, const double* __restrict__ K, const double* __restrict__ M, int ntab,
                                                     const double2* __restrict__ table, double tk, double tm,
                                                     uint8_t* __restrict__ code, int* __restrict__ fail, int cap,
                                                     double2* __restrict__ fail_v) {
  __shared__ double2 st[256];
  if ((int)threadIdx.x < ntab) st[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  // (tk, tm) are powers of two: the pair an entry is replaced by is ITS OWN value rounded to that grid - a function of the entry
  // alone, so the operator is bitwise the same whichever order the table was filled in
  const double kv = rint(K[k] / tk) * tk, mv = rint(M[k] / tm) * tm;
  int found = -1;
  for (int t = 0; t < ntab; ++t)
    if (kv == st[t].x && mv == st[t].y) {
      found = t;
      break;
    }
  if (found >= 0) {
    code[k] = (uint8_t)found;
    return;
  }
  if (*(volatile int*)fail < cap) {  // racy on purpose: it only bounds the number of atomics
    const int q = atomicAdd(fail, 1);
    if (q < cap) fail_v[q] = make_double2(kv, mv);
  } else {
    *(volatile int*)(fail + 1) = 1;  // "more than cap"
  }
}

void pgxk_dict_assign(hipStream_t st, int64_t nnz, const double* K, const double* M, int ntab, const double* table, double tk,
                      double tm, uint8_t* code, int* fail, int cap, double* fail_v) {
  hipLaunchKernelGGL(k_dict_assign, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, K, M, ntab, (const double2*)table, tk,
                     tm, code, fail, cap, (double2*)fail_v);
}

void pgxk_bspmv_stream(hipStream_t st, int n, size_t fill_lds_bytes, const int32_t* rowptr, const int32_t* colm,
                       const double* K, const double* M, const double* D, double alpha, const uint8_t* mask,
                       const double* xu, const double* xp, int remap, double* yu, double* yp) {
  const int cap = (int)(fill_lds_bytes / sizeof(double));
  if (first_use_on_device(0))  // P2 rows (up to 19 nnz) need more than the default 64 KB of dynamic LDS
    hipFuncSetAttribute((const void*)k_bspmv_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(k_bspmv_stream, dim3((n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK),
                     2 * fill_lds_bytes, st, n, cap, rowptr, colm, K, M, D, alpha, mask, xu, xp, remap, yu, yp);
}

void pgxk_bspmv(hipStream_t st, int mode, int n, const int32_t* rowptr, const int32_t* colm, const double* K,
                const double* M, const double* D, double alpha, const double* xu, const double* xp, const double* bu,
                const double* bp, double omega, int first, double* yu, double* yp) {
  constexpr int LPR = 8;
  const size_t threads = (size_t)n * LPR;
  dim3 grid((unsigned)((threads + PGX_BLOCK - 1) / PGX_BLOCK)), block(PGX_BLOCK);
  if (mode == 0)
    hipLaunchKernelGGL((k_bspmv<0, LPR>), grid, block, 0, st, n, rowptr, colm, K, M, D, alpha, xu, xp, bu, bp, omega,
                       first, yu, yp);
  else if (mode == 1)
    hipLaunchKernelGGL((k_bspmv<1, LPR>), grid, block, 0, st, n, rowptr, colm, K, M, D, alpha, xu, xp, bu, bp, omega,
                       first, yu, yp);
  else
    hipLaunchKernelGGL((k_bspmv<2, LPR>), grid, block, 0, st, n, rowptr, colm, K, M, D, alpha, xu, xp, bu, bp, omega,
                       first, yu, yp);
}

// ------------------------------------------------------------------------------------------------
// six observables of obstacle_pg.py:145-152 in ONE pass over the cells (the reference makes six)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PGX_BLOCK) k_observables(int nc, int n, const int32_t* __restrict__ cells,
                                                           const double* __restrict__ coords,
                                                           const double* __restrict__ x,
                                                           const double* __restrict__ xk, double alpha, double f,
                                                           QuadTab q, double* __restrict__ partials) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    const int v[3] = {cells[3 * c], cells[3 * c + 1], cells[3 * c + 2]};
    double u[3], p[3], uk[3], pk[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[a] = x[v[a]];
      p[a] = x[n + v[a]];
      uk[a] = xk[v[a]];
      pk[a] = xk[n + v[a]];
    }
    const P1Geom g = p1_geom(coords[2 * v[0]], coords[2 * v[0] + 1], coords[2 * v[1]], coords[2 * v[1] + 1],
                             coords[2 * v[2]], coords[2 * v[2] + 1]);
    const double area = 0.5 * g.adet;
    const double gx = u[0] * g.G[0][0] + u[1] * g.G[1][0] + u[2] * g.G[2][0];
    const double gy = u[0] * g.G[0][1] + u[1] * g.G[1][1] + u[2] * g.G[2][1];
    const double hx = gx - (uk[0] * g.G[0][0] + uk[1] * g.G[1][0] + uk[2] * g.G[2][0]);
    const double hy = gy - (uk[0] * g.G[0][1] + uk[1] * g.G[1][1] + uk[2] * g.G[2][1]);
    s[0] += 0.5 * area * (gx * gx + gy * gy);
    s[4] += area * (hx * hx + hy * hy);
    for (int k = 0; k < q.nq; ++k) {
      const double wd = g.adet * q.w[k];
      const double uq = u[0] * q.N[k][0] + u[1] * q.N[k][1] + u[2] * q.N[k][2];
      const double pq = p[0] * q.N[k][0] + p[1] * q.N[k][1] + p[2] * q.N[k][2];
      const double ukq = uk[0] * q.N[k][0] + uk[1] * q.N[k][1] + uk[2] * q.N[k][2];
      const double pkq = pk[0] * q.N[k][0] + pk[1] * q.N[k][1] + pk[2] * q.N[k][2];
      s[0] -= f * wd * uq;
      s[1] += wd * (pkq - pq) / alpha * uq;
      s[2] += wd * (uq < 0.0 ? -uq : 0.0);
      s[3] += wd * (pkq < pq ? (pq - pkq) / alpha : 0.0);
      const double du = uq - ukq;
      s[4] += wd * du * du;
      const double de = exp(pq) - exp(pkq);
      s[5] += wd * de * de;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double r = block_sum(s[k], sm);
    if (threadIdx.x == 0) partials[blockIdx.x * 6 + k] = r;
  }
}

// raw != 0: plain sums (sharded path: the all-reduce comes before abs / sqrt)
__global__ void __launch_bounds__(PGX_BLOCK) k_observables_final(int nblocks, const double* __restrict__ partials,
                                                                 double* __restrict__ out6, int raw) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  for (int k = 0; k < 6; ++k) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[b * 6 + k];
    const double r = block_sum(s, sm);
    if (threadIdx.x == 0) {
      double v = r;
      if (k == 1 && !raw) v = fabs(v);
      if (k >= 4 && !raw) v = sqrt(v);
      out6[k] = v;
    }
  }
}

void pgxk_observables_final(hipStream_t st, int nblocks, const double* partials, double* out6) {
  hipLaunchKernelGGL(k_observables_final, dim3(1), dim3(PGX_BLOCK), 0, st, nblocks, partials, out6, 0);
}
void pgxk_observables_final_raw(hipStream_t st, int nblocks, const double* partials, double* out6) {
  hipLaunchKernelGGL(k_observables_final, dim3(1), dim3(PGX_BLOCK), 0, st, nblocks, partials, out6, 1);
}

int pgxk_observables_blocks(int nc) {
  int b = (nc + PGX_BLOCK - 1) / PGX_BLOCK;
  return b < 2048 ? b : 2048;
}

void pgxk_observables(hipStream_t st, int nc, int n, const int32_t* cells, const double* coords, const double* x,
                      const double* xk, double alpha, double f, QuadTab q, double* partials, int nblocks,
                      double* out6, int raw, const double* geo) {
  if (geo)
    hipLaunchKernelGGL(k_observables_c, dim3(nblocks), dim3(PGX_BLOCK), 0, st, nc, n, cells, x, xk, alpha, f, q, partials, geo);
  else
    hipLaunchKernelGGL(k_observables, dim3(nblocks), dim3(PGX_BLOCK), 0, st, nc, n, cells, coords, x, xk, alpha, f, q,
                       partials);
  hipLaunchKernelGGL(k_observables_final, dim3(1), dim3(PGX_BLOCK), 0, st, nblocks, partials, out6, raw);
}

// ------------------------------------------------------------------------------------------------
// vector kernels (16 B / lane accesses; len is even and all bases 16-B aligned by construction)
// ------------------------------------------------------------------------------------------------
static inline dim3 stream_grid(size_t len2) {
  size_t b = (len2 + PGX_BLOCK - 1) / PGX_BLOCK;
  if (b > 4096) b = 4096;
  if (b == 0) b = 1;
  return dim3((unsigned)b);
}

__global__ void __launch_bounds__(PGX_BLOCK) k_axpy(size_t len2, double a, const double2* __restrict__ x,
                                                    double2* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 xv = x[i], yv = y[i];
    yv.x += a * xv.x;
    yv.y += a * xv.y;
    y[i] = yv;
  }
}
void pgxk_axpy(hipStream_t st, size_t len, double a, const double* x, double* y) {
  hipLaunchKernelGGL(k_axpy, stream_grid(len / 2), dim3(PGX_BLOCK), 0, st, len / 2, a, (const double2*)x,
                     (double2*)y);
}

__global__ void __launch_bounds__(PGX_BLOCK) k_scale_copy(size_t len2, double a, const double2* __restrict__ x,
                                                          double2* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 xv = x[i];
    xv.x *= a;
    xv.y *= a;
    y[i] = xv;
  }
}
void pgxk_scale_copy(hipStream_t st, size_t len, double a, const double* x, double* y) {
  hipLaunchKernelGGL(k_scale_copy, stream_grid(len / 2), dim3(PGX_BLOCK), 0, st, len / 2, a, (const double2*)x,
                     (double2*)y);
}

__global__ void __launch_bounds__(PGX_BLOCK) k_set(size_t len, double a, double* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a;
}
void pgxk_set(hipStream_t st, size_t len, double a, double* y) {
  hipLaunchKernelGGL(k_set, stream_grid(len), dim3(PGX_BLOCK), 0, st, len, a, y);
}

__global__ void __launch_bounds__(PGX_BLOCK) k_to_float(size_t len, const double* __restrict__ x, float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
void pgxk_to_float(hipStream_t st, size_t len, const double* x, float* y) {
  hipLaunchKernelGGL(k_to_float, stream_grid(len), dim3(PGX_BLOCK), 0, st, len, x, y);
}

// out[i] = V_i . w : each block streams a slice of w ONCE for NV vectors (Gram-Schmidt is the
// second-largest HBM consumer of the Newton solve; batching cuts its traffic from 2 to 1+1/NV
// vector reads per dot product).  Two-stage, fixed grid -> bitwise reproducible.
template <int NV>
__global__ void __launch_bounds__(PGX_BLOCK) k_multidot(size_t len2, const double2* __restrict__ V, size_t ldv2,
                                                        const double2* __restrict__ w,
                                                        double* __restrict__ partials, int pstride, int poff) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    const double2 wv = w[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double2 a = ldnt2(V + v * ldv2 + i);
      acc[v] += a.x * wv.x + a.y * wv.y;
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double r = block_sum(acc[v], sm);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * pstride + poff + v] = r;
  }
}

__global__ void __launch_bounds__(PGX_BLOCK) k_reduce_partials(int nblocks, int nv, const double* __restrict__ p,
                                                               double* __restrict__ out) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  const int v = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += p[(size_t)b * nv + v];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) out[v] = r;
}
// the same with a factor per dot product (lazy normalisation of the Krylov basis: pgx_api.hip, fgmres)
__global__ void __launch_bounds__(PGX_BLOCK) k_reduce_partials_scaled(int nblocks, int nv, const double* __restrict__ p, PgxDotScale sc,
                                                                      double* __restrict__ out) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  const int v = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += p[(size_t)b * nv + v];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) out[v] = r * sc.s[v];
}

void pgxk_multidot(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* w, double* partials,
                   double* out, const PgxDotScale* scale) {
  const size_t len2 = len / 2, ldv2 = ldv / 2;
  size_t nb = (len2 + PGX_BLOCK - 1) / PGX_BLOCK;
  if (nb > PGX_RED_BLOCKS) nb = PGX_RED_BLOCKS;
  if (nb == 0) nb = 1;
  dim3 grid((unsigned)nb), block(PGX_BLOCK);
  int done = 0;
  while (done < nv) {
    const int rem = nv - done;
    const double2* Vp = (const double2*)(V + (size_t)done * ldv);
    // one launch per chunk of at most 8 vectors, the last chunk of EXACTLY the remaining size: w is read once per chunk (the
    // 8/4/2/1 split of rounds 1-3 read it up to three times); the per-vector partials do not depend on the grouping
#define PGX_MD(N)                                                                                                          \
  case N:                                                                                                                  \
    hipLaunchKernelGGL(k_multidot<N>, grid, block, 0, st, len2, Vp, ldv2, (const double2*)w, partials, nv, done);          \
    break
    const int take = rem >= 8 ? 8 : rem;
    switch (take) {
      PGX_MD(1);
      PGX_MD(2);
      PGX_MD(3);
      PGX_MD(4);
      PGX_MD(5);
      PGX_MD(6);
      PGX_MD(7);
      PGX_MD(8);
    }
#undef PGX_MD
    done += take;
  }
  if (scale)
    hipLaunchKernelGGL(k_reduce_partials_scaled, dim3(nv), block, 0, st, (int)nb, nv, partials, *scale, out);
  else
    hipLaunchKernelGGL(k_reduce_partials, dim3(nv), block, 0, st, (int)nb, nv, partials, out);
}

template <int NV>
__global__ void __launch_bounds__(PGX_BLOCK) k_multiaxpy(size_t len2, const double2* __restrict__ V, size_t ldv2,
                                                         const double* __restrict__ h, double2* __restrict__ w) {
  double hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = h[v];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 wv = w[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double2 a = V[v * ldv2 + i];
      wv.x -= hv[v] * a.x;
      wv.y -= hv[v] * a.y;
    }
    w[i] = wv;
  }
}

// CGS2, passes 2+3 fused:  w' = w - V h1  and then  [h2; |w'|^2] = [V, w']^T w'  in ONE pass over the basis.
// A block owns a 256-element slice of every basis vector: while it forms w' it parks the slices in LDS
// (nv x 2 KB), then dots them against w' from LDS - the basis is read from HBM once instead of twice.
// Two-stage fixed-shape reduction (per-block partials, then k_reduce_partials) -> bitwise reproducible.
__global__ void __launch_bounds__(PGX_BLOCK) k_axpy_dot(size_t len, int nv, const double* __restrict__ V, size_t ldv,
                                                        const double* __restrict__ h1, double* __restrict__ w,
                                                        double* __restrict__ partials) {
  extern __shared__ double sh[];  // [nv+1][256] slices (+ w'), then hs[nv]
  double* hs = sh + (size_t)(nv + 1) * PGX_BLOCK;
  const int t = threadIdx.x;
  const int lane = t & (WAVE - 1), wid = t / WAVE;
  if (t < nv) hs[t] = h1[t];
  constexpr int NW = PGX_BLOCK / WAVE;
  constexpr int MAXA = 16;  // ceil(62 / 4): vectors handled by one wave
  double acc[MAXA];
#pragma unroll
  for (int k = 0; k < MAXA; ++k) acc[k] = 0.0;
  const size_t nchunks = (len + PGX_BLOCK - 1) / PGX_BLOCK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    __syncthreads();  // previous chunk's LDS image fully consumed (also orders the hs[] fill the first time)
    const size_t i = chunk * PGX_BLOCK + t;
    const bool live = i < len;
    double wv = live ? w[i] : 0.0;
#pragma unroll 8
    for (int v = 0; v < nv; ++v) {
      const double a = live ? __builtin_nontemporal_load(V + (size_t)v * ldv + i) : 0.0;
      sh[v * PGX_BLOCK + t] = a;
      wv -= hs[v] * a;
    }
    if (live) w[i] = wv;
    sh[nv * PGX_BLOCK + t] = wv;
    __syncthreads();
    const double* wl = sh + (size_t)nv * PGX_BLOCK;
    double wr[NW];
#pragma unroll
    for (int r = 0; r < NW; ++r) wr[r] = wl[lane + WAVE * r];
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
      const int v = wid + NW * k;
      if (v <= nv) {
        const double* sv = sh + (size_t)v * PGX_BLOCK;
#pragma unroll
        for (int r = 0; r < NW; ++r) acc[k] += sv[lane + WAVE * r] * wr[r];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXA; ++k) {
    const int v = wid + NW * k;
    if (v <= nv) {
      const double r = wave_sum(acc[k]);
      if (lane == 0) partials[(size_t)v * gridDim.x + blockIdx.x] = r;  // [v][block]: coalesced second stage
    }
  }
}

// out[v] = sum_b p[v][b]  (row-contiguous partials)
__global__ void __launch_bounds__(PGX_BLOCK) k_reduce_rows(int nb, const double* __restrict__ p,
                                                           double* __restrict__ out) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  const double* row = p + (size_t)blockIdx.x * nb;
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s += row[b];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// out[0..nv-1] = h2, out[nv] = |w'|^2.  partials must hold ceil(len/256) * (nv+1) doubles.
void pgxk_axpy_dot(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h1, double* w,
                   double* partials, double* out) {
  size_t nchunks = (len + PGX_BLOCK - 1) / PGX_BLOCK;
  const unsigned nb = (unsigned)std::min<size_t>(nchunks, 4096);  // blocks loop over chunks: 8x fewer partials at 2048^2
  const size_t lds = ((size_t)(nv + 1) * PGX_BLOCK + nv) * sizeof(double);
  if (first_use_on_device(1))
    hipFuncSetAttribute((const void*)k_axpy_dot, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(k_axpy_dot, dim3(nb), dim3(PGX_BLOCK), lds, st, len, nv, V, ldv, h1, w, partials);
  hipLaunchKernelGGL(k_reduce_rows, dim3(nv + 1), dim3(PGX_BLOCK), 0, st, (int)nb, partials, out);
}

// CGS2 pass 4 fused with the normalisation:  v_next = (w - V h2) * scale
template <int NV>
__global__ void __launch_bounds__(PGX_BLOCK) k_multiaxpy_scale(size_t len2, const double2* __restrict__ V, size_t ldv2,
                                                               const double* __restrict__ h, double scale, int last,
                                                               double2* __restrict__ w) {
  double hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = h[v];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 wv = w[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double2 a = ldnt2(V + v * ldv2 + i);
      wv.x -= hv[v] * a.x;
      wv.y -= hv[v] * a.y;
    }
    if (last) {
      wv.x *= scale;
      wv.y *= scale;
    }
    w[i] = wv;
  }
}

void pgxk_multiaxpy_scale(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h,
                          double scale, double* w) {
  const size_t len2 = len / 2, ldv2 = ldv / 2;
  dim3 grid = stream_grid(len2), block(PGX_BLOCK);
  int done = 0;
  while (done < nv) {
    const int rem = nv - done;
    const double2* Vp = (const double2*)(V + (size_t)done * ldv);
    const int step = rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1;
    const int last = (done + step == nv);
    if (step == 8)
      hipLaunchKernelGGL(k_multiaxpy_scale<8>, grid, block, 0, st, len2, Vp, ldv2, h + done, scale, last, (double2*)w);
    else if (step == 4)
      hipLaunchKernelGGL(k_multiaxpy_scale<4>, grid, block, 0, st, len2, Vp, ldv2, h + done, scale, last, (double2*)w);
    else if (step == 2)
      hipLaunchKernelGGL(k_multiaxpy_scale<2>, grid, block, 0, st, len2, Vp, ldv2, h + done, scale, last, (double2*)w);
    else
      hipLaunchKernelGGL(k_multiaxpy_scale<1>, grid, block, 0, st, len2, Vp, ldv2, h + done, scale, last, (double2*)w);
    done += step;
  }
}

// w -= sum_v h[v] V_v in chunks of <= 8 vectors; the LAST chunk also leaves the block's share of |w'|^2 in partials[block]
// (fixed-shape two-stage reduction: reproducible).  The lean second pass of selective CGS2: one read of the basis, one
// read-modify-write of w, no LDS parking of basis slices (k_axpy_dot does that to get V^T w' in the same pass, which is only
// needed when the second projection is - 11 of 266 iterations at 2048^2).
template <int NV, bool NORM>
__global__ void __launch_bounds__(PGX_BLOCK) k_multiaxpy_norm(size_t len2, const double2* __restrict__ V, size_t ldv2,
                                                              const double* __restrict__ h, double2* __restrict__ w,
                                                              double* __restrict__ partials) {
  __shared__ double sm[PGX_BLOCK / WAVE];
  double hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = h[v];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 wv = w[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double2 a = ldnt2(V + v * ldv2 + i);
      wv.x -= hv[v] * a.x;
      wv.y -= hv[v] * a.y;
    }
    w[i] = wv;
    if (NORM) acc += wv.x * wv.x + wv.y * wv.y;
  }
  if (NORM) {
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
  }
}
template <int NV>
static void launch_multiaxpy_norm(hipStream_t st, dim3 grid, size_t len2, const double2* Vp, size_t ldv2, const double* h, bool norm,
                                  double2* w, double* partials) {
  if (norm)
    hipLaunchKernelGGL((k_multiaxpy_norm<NV, true>), grid, dim3(PGX_BLOCK), 0, st, len2, Vp, ldv2, h, w, partials);
  else
    hipLaunchKernelGGL((k_multiaxpy_norm<NV, false>), grid, dim3(PGX_BLOCK), 0, st, len2, Vp, ldv2, h, w, partials);
}
// out[0] = |w - V h|^2, w updated in place.  partials: >= PGX_RED_BLOCKS doubles.
void pgxk_multiaxpy_norm(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h, double* w,
                         double* partials, double* out) {
  const size_t len2 = len / 2, ldv2 = ldv / 2;
  dim3 grid = stream_grid(len2);
  int done = 0;
  while (done < nv) {
    const int step = std::min(8, nv - done);
    const bool last = done + step == nv;
    const double2* Vp = (const double2*)(V + (size_t)done * ldv);
    switch (step) {
      case 8: launch_multiaxpy_norm<8>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 7: launch_multiaxpy_norm<7>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 6: launch_multiaxpy_norm<6>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 5: launch_multiaxpy_norm<5>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 4: launch_multiaxpy_norm<4>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 3: launch_multiaxpy_norm<3>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      case 2: launch_multiaxpy_norm<2>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
      default: launch_multiaxpy_norm<1>(st, grid, len2, Vp, ldv2, h + done, last, (double2*)w, partials); break;
    }
    done += step;
  }
  hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(PGX_BLOCK), 0, st, (int)grid.x, partials, out);
}

void pgxk_multiaxpy(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h, double* w) {
  const size_t len2 = len / 2, ldv2 = ldv / 2;
  dim3 grid = stream_grid(len2), block(PGX_BLOCK);
  int done = 0;
  while (done < nv) {
    const int rem = nv - done;
    const double2* Vp = (const double2*)(V + (size_t)done * ldv);
    if (rem >= 8) {
      hipLaunchKernelGGL(k_multiaxpy<8>, grid, block, 0, st, len2, Vp, ldv2, h + done, (double2*)w);
      done += 8;
    } else if (rem >= 4) {
      hipLaunchKernelGGL(k_multiaxpy<4>, grid, block, 0, st, len2, Vp, ldv2, h + done, (double2*)w);
      done += 4;
    } else if (rem >= 2) {
      hipLaunchKernelGGL(k_multiaxpy<2>, grid, block, 0, st, len2, Vp, ldv2, h + done, (double2*)w);
      done += 2;
    } else {
      hipLaunchKernelGGL(k_multiaxpy<1>, grid, block, 0, st, len2, Vp, ldv2, h + done, (double2*)w);
      done += 1;
    }
  }
}

// x (+)= sum_i y[i] Z_i
__global__ void __launch_bounds__(PGX_BLOCK) k_lincomb(size_t len2, int nv, const double2* __restrict__ Z,
                                                       size_t ldz2, const double* __restrict__ y,
                                                       double2* __restrict__ x, int accumulate) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len2; i += (size_t)gridDim.x * blockDim.x) {
    double2 s = accumulate ? x[i] : make_double2(0.0, 0.0);
    for (int v = 0; v < nv; ++v) {
      const double2 a = Z[v * ldz2 + i];
      const double yv = y[v];
      s.x += yv * a.x;
      s.y += yv * a.y;
    }
    x[i] = s;
  }
}
// (xu, xp)[v] += sum_i y[i] Zf_i[v] for float2-interleaved Z_i (see st_load_x): the solution update of a cycle whose Z_j are float
__global__ void __launch_bounds__(PGX_BLOCK) k_lincomb_f2(size_t n, int nv, const float2* __restrict__ Zf, size_t ldz,
                                                          const double* __restrict__ y, double* __restrict__ xu, double* __restrict__ xp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double su = xu[i], sp = xp[i];
    for (int v = 0; v < nv; ++v) {
      const float2 a = Zf[v * ldz + i];
      const double yv = y[v];
      su += yv * (double)a.x;
      sp += yv * (double)a.y;
    }
    xu[i] = su;
    xp[i] = sp;
  }
}
void pgxk_lincomb_f2(hipStream_t st, size_t n, int nv, const float2* Zf, size_t ldz, const double* y, double* xu, double* xp) {
  const unsigned grid = (unsigned)std::min<size_t>((n + PGX_BLOCK - 1) / PGX_BLOCK, 256 * 32);
  hipLaunchKernelGGL(k_lincomb_f2, dim3(grid), dim3(PGX_BLOCK), 0, st, n, nv, Zf, ldz, y, xu, xp);
}

void pgxk_lincomb(hipStream_t st, size_t len, int nv, const double* Z, size_t ldz, const double* y, double* x,
                  int accumulate) {
  hipLaunchKernelGGL(k_lincomb, stream_grid(len / 2), dim3(PGX_BLOCK), 0, st, len / 2, nv, (const double2*)Z,
                     ldz / 2, y, (double2*)x, accumulate);
}

// ------------------------------------------------------------------------------------------------
// 7-point stencil multigrid (structured right-diagonal meshes, nested by vertex coarsening)
// ------------------------------------------------------------------------------------------------
// P1 prolongation weight of the fine vertex at offset (dx,dy) from a coarse vertex
__device__ __forceinline__ constexpr double pw(int dx, int dy) {
  return (dx == 0 && dy == 0) ? 1.0
         : ((dx == 1 && dy == 0) || (dx == -1 && dy == 0) || (dx == 0 && dy == 1) || (dx == 0 && dy == -1) ||
            (dx == 1 && dy == 1) || (dx == -1 && dy == -1))
             ? 0.5
             : 0.0;
}

__global__ void __launch_bounds__(PGX_BLOCK) k_csr_to_stencil(int n, int sx, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ colm,
                                                              const double* __restrict__ vals,
                                                              double* __restrict__ S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int o = (colm[k] & 0x7fffffff) - i;
    const double v = vals[k];
    if (o == 0) s[0] = v;
    else if (o == 1) s[1] = v;
    else if (o == -1) s[2] = v;
    else if (o == sx) s[3] = v;
    else if (o == -sx) s[4] = v;
    else if (o == sx + 1) s[5] = v;
    else if (o == -sx - 1) s[6] = v;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) S[(size_t)k * n + i] = s[k];
}
void pgxk_csr_to_stencil(hipStream_t st, int n, int sx, const int32_t* rowptr, const int32_t* colm, const double* vals,
                         double* S) {
  hipLaunchKernelGGL(k_csr_to_stencil, dim3((n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, n, sx, rowptr,
                     colm, vals, S);
}

// Galerkin coarse operator S_c = P^T S_f P for a 7-point stencil; stays 7-point because the P1 spaces
// are nested.  One thread per coarse vertex; all 7x7x7 index combinations are resolved at compile time.
__global__ void __launch_bounds__(PGX_BLOCK) k_rap7(int nxf, int nyf, int nf, const double* __restrict__ Sf, int nxc,
                                                    int nyc, int ncv, double* __restrict__ Sc) {
  const int C = blockIdx.x * blockDim.x + threadIdx.x;
  if (C >= ncv) return;
  constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
  constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
  const int sxc = nxc + 1, sxf = nxf + 1;
  const int I = C % sxc, Jc = C / sxc;
  double out[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int oa = 0; oa < 7; ++oa) {
    const int ax = 2 * I + OX[oa], ay = 2 * Jc + OY[oa];
    if (ax < 0 || ax > nxf || ay < 0 || ay > nyf) continue;
    const double wa = pw(OX[oa], OY[oa]);
    const size_t a = (size_t)ay * sxf + ax;
#pragma unroll
    for (int os = 0; os < 7; ++os) {
      const double coef = wa * Sf[(size_t)os * nf + a];
      const int dx = OX[oa] + OX[os], dy = OY[oa] + OY[os];
#pragma unroll
      for (int oc = 0; oc < 7; ++oc) {
        const double w2 = pw(dx - 2 * OX[oc], dy - 2 * OY[oc]);
        if (w2 != 0.0) out[oc] += coef * w2;
      }
    }
  }
#pragma unroll
  for (int oc = 0; oc < 7; ++oc) {
    const int nx_ = I + OX[oc], ny_ = Jc + OY[oc];
    const bool ok = nx_ >= 0 && nx_ <= nxc && ny_ >= 0 && ny_ <= nyc;
    Sc[(size_t)oc * ncv + C] = ok ? out[oc] : 0.0;
  }
}
void pgxk_rap7(hipStream_t st, const GridLevel& f, const double* Sf, const GridLevel& c, double* Sc) {
  hipLaunchKernelGGL(k_rap7, dim3((c.n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, f.nx, f.ny, f.n, Sf, c.nx,
                     c.ny, c.n, Sc);
}

// Load the 7 coefficients of a symmetric-half stencil at vertex v=(i,j): the negative-direction links
// are the neighbours' positive ones.  Sh slots: 0:(0,0) 1:(+1,0) 2:(0,+1) 3:(+1,+1).
__device__ __forceinline__ void ld7h(const dsten_t* __restrict__ Sh, int n, int sx, int v, int i, int j, int nx,
                                     int ny, double d[7]) {
  d[0] = Sh[v];
  d[1] = (i < nx) ? Sh[(size_t)n + v] : 0.0;
  d[2] = (i > 0) ? Sh[(size_t)n + v - 1] : 0.0;
  d[3] = (j < ny) ? Sh[(size_t)2 * n + v] : 0.0;
  d[4] = (j > 0) ? Sh[(size_t)2 * n + v - sx] : 0.0;
  d[5] = (i < nx && j < ny) ? Sh[(size_t)3 * n + v] : 0.0;
  d[6] = (i > 0 && j > 0) ? Sh[(size_t)3 * n + v - sx - 1] : 0.0;
}

__global__ void __launch_bounds__(PGX_BLOCK) k_csr_to_stencil_h(int n, int sx, const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ colm,
                                                                const double* __restrict__ vals,
                                                                dsten_t* __restrict__ Sh, int frame_ny) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (frame_ny > 0) {  // boundary frame only: k_resid_fill_grid has written the interior rows
    const int gi = i % sx, gj = i / sx;
    if (gi > 0 && gi < sx - 1 && gj > 0 && gj < frame_ny) return;
  }
  double s[4] = {0, 0, 0, 0};
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int o = (colm[k] & 0x7fffffff) - i;
    const double v = vals[k];
    if (o == 0) s[0] = v;
    else if (o == 1) s[1] = v;
    else if (o == sx) s[2] = v;
    else if (o == sx + 1) s[3] = v;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) Sh[(size_t)k * n + i] = (dsten_t)s[k];
}
void pgxk_csr_to_stencil_h(hipStream_t st, int n, int sx, const int32_t* rowptr, const int32_t* colm,
                           const double* vals, dsten_t* Sh, int frame_ny) {
  hipLaunchKernelGGL(k_csr_to_stencil_h, dim3((n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, n, sx, rowptr,
                     colm, vals, Sh, frame_ny);
}

// Galerkin coarsening on symmetric-half storage: only the 4 stored coarse slots are produced.
__global__ void __launch_bounds__(PGX_BLOCK) k_rap7h(int nxf, int nyf, int nf, const dsten_t* __restrict__ Sfh, int nxc,
                                                     int nyc, int ncv, dsten_t* __restrict__ Sch) {
  const int C = blockIdx.x * blockDim.x + threadIdx.x;
  if (C >= ncv) return;
  constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
  constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
  constexpr int KEEP[4] = {0, 1, 3, 5};  // full-stencil slots stored in the half format
  const int sxc = nxc + 1, sxf = nxf + 1;
  const int I = C % sxc, Jc = C / sxc;
  double out[4] = {0, 0, 0, 0};
#pragma unroll
  for (int oa = 0; oa < 7; ++oa) {
    const int ax = 2 * I + OX[oa], ay = 2 * Jc + OY[oa];
    if (ax < 0 || ax > nxf || ay < 0 || ay > nyf) continue;
    const double wa = pw(OX[oa], OY[oa]);
    const int a = ay * sxf + ax;
    double cf[7];
    ld7h(Sfh, nf, sxf, a, ax, ay, nxf, nyf, cf);
#pragma unroll
    for (int os = 0; os < 7; ++os) {
      const double coef = wa * cf[os];
      const int dx = OX[oa] + OX[os], dy = OY[oa] + OY[os];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double w2 = pw(dx - 2 * OX[KEEP[q]], dy - 2 * OY[KEEP[q]]);
        if (w2 != 0.0) out[q] += coef * w2;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nx_ = I + OX[KEEP[q]], ny_ = Jc + OY[KEEP[q]];
    const bool ok = nx_ >= 0 && nx_ <= nxc && ny_ >= 0 && ny_ <= nyc;
    Sch[(size_t)q * ncv + C] = (dsten_t)(ok ? out[q] : 0.0);
  }
}
void pgxk_rap7h(hipStream_t st, const GridLevel& f, const dsten_t* Sfh, const GridLevel& c, dsten_t* Sch) {
  hipLaunchKernelGGL(k_rap7h, dim3((c.n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, f.nx, f.ny, f.n, Sfh,
                     c.nx, c.ny, c.n, Sch);
}

__global__ void k_coarse_mask(int nxc, int nyc, int ncv, int nxf, const uint8_t* __restrict__ mf,
                              uint8_t* __restrict__ mc) {
  const int C = blockIdx.x * blockDim.x + threadIdx.x;
  if (C >= ncv) return;
  const int I = C % (nxc + 1), J = C / (nxc + 1);
  mc[C] = mf[(size_t)(2 * J) * (nxf + 1) + 2 * I];
}
void pgxk_coarse_mask(hipStream_t st, const GridLevel& c, uint8_t* mask_c, const GridLevel& f) {
  hipLaunchKernelGGL(k_coarse_mask, dim3((c.n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, c.nx, c.ny, c.n,
                     f.nx, f.mask, mask_c);
}

// Sharded path: ns SoA arrays computed on this rank's strip view of a replicated level -> the level's global arrays,
// owned rows copied, every other row 0 (the all-reduce that follows then assembles the level exactly: one non-zero
// contribution per entry).
__global__ void __launch_bounds__(PGX_BLOCK) k_view_to_global(int ns, int n_view, int n_glob, int sx, int row0,
                                                              int own0, int nown, const double* __restrict__ in,
                                                              double* __restrict__ out) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= (size_t)ns * n_glob) return;
  const int s = (int)(t / n_glob), v = (int)(t % n_glob);
  const int J = v / sx;
  out[t] = (J >= own0 && J < own0 + nown) ? in[(size_t)s * n_view + (v - row0 * sx)] : 0.0;
}
void pgxk_view_to_global(hipStream_t st, int ns, int n_view, int n_glob, int sx, int row0, int own0, int nown,
                         const double* in, double* out) {
  const size_t tot = (size_t)ns * n_glob;
  hipLaunchKernelGGL(k_view_to_global, dim3((unsigned)((tot + PGX_BLOCK - 1) / PGX_BLOCK)), dim3(PGX_BLOCK), 0, st, ns,
                     n_view, n_glob, sx, row0, own0, nown, in, out);
}

// One vertex of the collective operator / smoother on stencil storage (same three modes as k_bspmv).
// Per vertex this streams 4 D coefficients + the vectors (~80 B) instead of a CSR row (~250 B): on a
// uniform grid the K and M stencils of interior vertices are kernel-argument constants, only boundary
// vertices read arrays.
template <int MODE>
__device__ __forceinline__ void st_vertex(int v, int nx, int ny, int n, const double* __restrict__ K,
                                          const double* __restrict__ M, const dsten_t* __restrict__ Dh,
                                          const StConst& sc, const uint8_t* __restrict__ mask, double alpha,
                                          const double* xu, const double* xp, const double* bu, const double* bp,
                                          double omega, int first, double* yu, double* yp) {
  const int sx = nx + 1;
  const int i = v % sx, j = v / sx;
  const int off[7] = {0, 1, -1, sx, -sx, sx + 1, -sx - 1};
  const bool ok[7] = {true, i < nx, i > 0, j < ny, j > 0, i < nx && j < ny, i > 0 && j > 0};
  const bool interior = sc.uniform && i > 0 && i < nx && j > 0 && j < ny;
  double kv[7], mv[7], dv[7];
  if (interior) {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      kv[s] = sc.K[s];
      mv[s] = sc.M[s];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      kv[s] = K[(size_t)s * n + v];
      mv[s] = M[(size_t)s * n + v];
    }
  }
  ld7h(Dh, n, sx, v, i, j, nx, ny, dv);
  const int rowbc = mask[v];
  double au = 0.0, ap = 0.0;
  const bool skip = (MODE == 2) && (first & 1);
  if (!skip) {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      if (!ok[s]) continue;
      const int nb = v + off[s];
      const double xuv = mask[nb] ? 0.0 : xu[nb];
      const double xpv = xp[nb];
      au += alpha * kv[s] * xuv + mv[s] * xpv;
      ap += mv[s] * xuv - dv[s] * xpv;
    }
  }
  const double xur = skip ? 0.0 : xu[v];
  if (rowbc) au = xur;
  if (MODE == 0) {
    yu[v] = au;
    yp[v] = ap;
  } else if (MODE == 1) {
    yu[v] = bu[v] - au;
    yp[v] = bp[v] - ap;
  } else {
    const double su = bu[v] - au, sp = bp[v] - ap;
    const double xpr = skip ? 0.0 : xp[v];
    double a = alpha * kv[0], b = mv[0];
    const double dd = dv[0];
    double om_u = omega;
    if (rowbc) {
      a = 1.0;
      b = 0.0;
      om_u = 1.0;
    }
    const double det = -a * dd - b * b;
    double du = 0.0, dpsi = 0.0;
    if (det != 0.0) {
      du = (-dd * su - b * sp) / det;
      dpsi = (-b * su + a * sp) / det;
    } else if (rowbc) {
      du = su;
    }
    yu[v] = xur + om_u * du;
    yp[v] = xpr + omega * dpsi;
  }
}

template <int MODE>
__global__ void __launch_bounds__(PGX_BLOCK) k_st_apply(int nx, int ny, int n, const double* __restrict__ K,
                                                        const double* __restrict__ M,
                                                        const dsten_t* __restrict__ Dh, StConst sc,
                                                        const uint8_t* __restrict__ mask, double alpha,
                                                        const double* __restrict__ xu, const double* __restrict__ xp,
                                                        const double* __restrict__ bu, const double* __restrict__ bp,
                                                        double omega, int first, double* __restrict__ yu,
                                                        double* __restrict__ yp) {
  const int v = xcd_block(blockIdx.x, gridDim.x, first >> 1) * blockDim.x + threadIdx.x;
  if (v >= n) return;
  st_vertex<MODE>(v, nx, ny, n, K, M, Dh, sc, mask, alpha, xu, xp, bu, bp, omega, first, yu, yp);
}
// ------------------------------------------------------------------------------------------------
// Fused V-cycle legs (nu = 2).  rocprof showed the cycle bound by (a) four full passes over the level
// for the four Jacobi sweeps and (b) ~8 dependent launches per level.  Here a level costs 3 launches:
//   k_st_smooth2<PRE>   x1 = S(S(0))                      (S = one collective damped-Jacobi sweep)
//   k_st_resid_restrict b_c = P^T (b - J x1)              (no residual round trip through HBM)
//   k_st_smooth2<POST>  x2 = S(S(x1 + P x_c))             (prolongation folded in)
// smooth2 works on 2-D tiles: sweep 1 is evaluated on the tile plus a one-vertex halo into LDS (redundant
// work (TX+2)(TY+2)/(TX*TY) = 1.16 for 64x16), sweep 2 reads it from LDS, so both sweeps cost ONE pass over
// D, b and x instead of two.
// ------------------------------------------------------------------------------------------------
struct StCoef {
  double kv[7], mv[7], dv[7];
  bool ok[7];
  int rowbc;
};

__device__ __forceinline__ void st_load_coef(int v, int i, int j, int nx, int ny, int n, const double* __restrict__ K,
                                             const double* __restrict__ M, const dsten_t* __restrict__ Dh,
                                             const StConst& sc, const uint8_t* __restrict__ mask, StCoef& c) {
  const int sx = nx + 1;
  c.ok[0] = true;
  c.ok[1] = i < nx;
  c.ok[2] = i > 0;
  c.ok[3] = j < ny;
  c.ok[4] = j > 0;
  c.ok[5] = i < nx && j < ny;
  c.ok[6] = i > 0 && j > 0;
  if (sc.uniform && i > 0 && i < nx && j > 0 && j < ny) {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      c.kv[s] = sc.K[s];
      c.mv[s] = sc.M[s];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      c.kv[s] = K[(size_t)s * n + v];
      c.mv[s] = M[(size_t)s * n + v];
    }
  }
  ld7h(Dh, n, sx, v, i, j, nx, ny, c.dv);
  c.rowbc = mask[v];
}

// (au, ap) = rows of J applied to neighbour values (xun must already be 0 at Dirichlet columns)
__device__ __forceinline__ void st_rows(const StCoef& c, double alpha, const double xun[7], const double xpn[7],
                                        double& au, double& ap) {
  au = 0.0;
  ap = 0.0;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    if (!c.ok[s]) continue;
    au += alpha * c.kv[s] * xun[s] + c.mv[s] * xpn[s];
    ap += c.mv[s] * xun[s] - c.dv[s] * xpn[s];
  }
}

// LDS-image variants for the fused sweep kernels: no per-link validity tests.  Out-of-grid halo entries of the image
// are 0 and Dirichlet entries of u are stored as 0 (pre-masked), and every link that leaves the grid has a zero
// coefficient (interior constants only apply to interior vertices), so the plain 7-term sums are exact.
__device__ __forceinline__ void st_rows_img(const StCoef& c, double alpha, const double* __restrict__ iu,
                                            const double* __restrict__ ip, int q0, int W, double& au, double& ap) {
  const int off[7] = {0, 1, -1, W, -W, W + 1, -W - 1};
  au = 0.0;
  ap = 0.0;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const double xu = iu[q0 + off[s]], xp = ip[q0 + off[s]];
    au += alpha * c.kv[s] * xu + c.mv[s] * xp;
    ap += c.mv[s] * xu - c.dv[s] * xp;
  }
}

__device__ __forceinline__ void st_jacobi(const StCoef& c, double alpha, double omega, double au, double ap,
                                          double xur, double xpr, double buv, double bpv, double& yu, double& yp) {
  if (c.rowbc) au = xur;
  const double su = buv - au, sp = bpv - ap;
  double a = alpha * c.kv[0], b = c.mv[0];
  const double dd = c.dv[0];
  double om_u = omega;
  if (c.rowbc) {
    a = 1.0;
    b = 0.0;
    om_u = 1.0;
  }
  const double det = -a * dd - b * b;
  double du = 0.0, dpsi = 0.0;
  if (det != 0.0) {
    // one hardware reciprocal + one Newton step instead of two IEEE divisions (~30 instructions): the smoother
    // is issue-bound, and this is a preconditioner - 1-2 ulp in 1/det cannot matter
    double r = __builtin_amdgcn_rcp(det);
    r = r * (2.0 - det * r);
    du = (-dd * su - b * sp) * r;
    dpsi = (-b * su + a * sp) * r;
  } else if (c.rowbc) {
    du = su;
  }
  yu = xur + om_u * du;
  yp = xpr + omega * dpsi;
}

template <int TX, int TY, bool POST>
__global__ void __launch_bounds__(PGX_BLOCK) k_st_smooth2(int nx, int ny, int n, const double* __restrict__ K,
                                                          const double* __restrict__ M,
                                                          const dsten_t* __restrict__ Dh, StConst sc,
                                                          const uint8_t* __restrict__ mask, double alpha,
                                                          const double* __restrict__ xu, const double* __restrict__ xp,
                                                          const double* __restrict__ cu, const double* __restrict__ cp,
                                                          int nxc, const double* __restrict__ bu,
                                                          const double* __restrict__ bp, double omega, int remap,
                                                          double* __restrict__ yu, double* __restrict__ yp) {
  constexpr int W2 = TX + 4, H2 = TY + 4, W1 = TX + 2, H1 = TY + 2;
  __shared__ double s0u[POST ? W2 * H2 : 1], s0p[POST ? W2 * H2 : 1];
  __shared__ double s1u[W1 * H1], s1p[W1 * H1];
  const int sx = nx + 1;
  const int ntx = (nx + TX) / TX;  // ceil((nx+1)/TX)
  const int b = xcd_block(blockIdx.x, gridDim.x, remap);
  const int i0 = (b % ntx) * TX, j0 = (b / ntx) * TY;
  const int tid = threadIdx.x;
  // phase 0 (POST): the corrected iterate x + P x_c on the tile + 2-halo
  for (int p = tid; POST && p < W2 * H2; p += PGX_BLOCK) {
    const int gi = i0 - 2 + p % W2, gj = j0 - 2 + p / W2;
    const bool in = gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
    const int v = gj * sx + gi;
    if (POST) {
      double a = 0.0, c2 = 0.0;
      if (in) {
        const int sxc = nxc + 1;
        const int ic = gi >> 1, jc = gj >> 1;
        const int c0 = jc * sxc + ic, c1 = (jc + (gj & 1)) * sxc + (ic + (gi & 1));
        a = xu[v];
        c2 = xp[v];
        if (cu) {  // cu == nullptr: plain double sweep of the current iterate (no coarse correction to add)
          a += 0.5 * (cu[c0] + cu[c1]);
          c2 += 0.5 * (cp[c0] + cp[c1]);
        }
      }
      s0u[p] = (in && mask[v]) ? 0.0 : a;  // pre-masked image: Dirichlet entries of u read as 0 by neighbours
      s0p[p] = c2;
    }
  }
  __syncthreads();
  // phase 1: sweep 1 on the tile + 1-halo -> LDS
  for (int p = tid; p < W1 * H1; p += PGX_BLOCK) {
    const int li = p % W1, lj = p / W1;
    const int gi = i0 - 1 + li, gj = j0 - 1 + lj;
    double r1u = 0.0, r1p = 0.0;
    if (gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
      const int v = gj * sx + gi;
      StCoef c;
      st_load_coef(v, gi, gj, nx, ny, n, K, M, Dh, sc, mask, c);
      double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
      if (POST) {
        const int q0 = (lj + 1) * W2 + (li + 1);
        st_rows_img(c, alpha, s0u, s0p, q0, W2, au, ap);
        xur = s0u[q0];
        xpr = s0p[q0];
      }
      st_jacobi(c, alpha, omega, au, ap, xur, xpr, bu[v], bp[v], r1u, r1p);
      if (c.rowbc) r1u = 0.0;  // pre-masked image (the bc row's own update never reads it: yu = bu)
    }
    s1u[p] = r1u;
    s1p[p] = r1p;
  }
  __syncthreads();
  // phase 2: sweep 2 on the tile, neighbours from LDS
  for (int p = tid; p < TX * TY; p += PGX_BLOCK) {
    const int li = p % TX, lj = p / TX;
    const int gi = i0 + li, gj = j0 + lj;
    if (gi > nx || gj > ny) continue;
    const int v = gj * sx + gi;
    StCoef c;
    st_load_coef(v, gi, gj, nx, ny, n, K, M, Dh, sc, mask, c);
    double au, ap, ou, op;
    const int q0 = (lj + 1) * W1 + (li + 1);
    st_rows_img(c, alpha, s1u, s1p, q0, W1, au, ap);
    st_jacobi(c, alpha, omega, au, ap, s1u[q0], s1p[q0], bu[v], bp[v], ou, op);
    yu[v] = ou;
    yp[v] = op;
  }
}

// post=0: (yu,yp) = S(S(0));  post=1: (yu,yp) = S(S((xu,xp) + P (cu,cp)))   -- out of place
void pgxk_st_smooth2(hipStream_t st, int post, const GridLevel& L, double alpha, const double* xu, const double* xp,
                     const GridLevel* C, const double* cu, const double* cp, const double* bu, const double* bp,
                     double omega, int remap, double* yu, double* yp) {
  constexpr int TX = PGX_TILE_X, TY = PGX_TILE_Y;
  const int ntx = (L.nx + TX) / TX, nty = (L.ny + TY) / TY;
  dim3 grid(ntx * nty), block(PGX_BLOCK);
  const StConst sc = make_stconst(L);
  if (post)
    hipLaunchKernelGGL((k_st_smooth2<TX, TY, true>), grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask,
                       alpha, xu, xp, cu, cp, C ? C->nx : 0, bu, bp, omega, remap, yu, yp);
  else
    hipLaunchKernelGGL((k_st_smooth2<TX, TY, false>), grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask,
                       alpha, nullptr, nullptr, nullptr, nullptr, 0, bu, bp, omega, remap, yu, yp);
}

// K sweeps per launch (generalisation of k_st_smooth2; K=3 serves the default nu=6 with ONE pass over the level
// per leg).  The iterate lives in an LDS ping-pong image of the tile + K-vertex halo; sweep s is evaluated on the
// tile + (K-s) halo, the last sweep writes the tile to HBM.  Redundant work for 64x16, K=3: 1.33 / 1.16 / 1.0.
template <int TX, int TY, int K, bool POST>
__global__ void __launch_bounds__(PGX_BLOCK) k_st_smoothK(int nx, int ny, int n, const double* __restrict__ Kc,
                                                          const double* __restrict__ M,
                                                          const dsten_t* __restrict__ Dh, StConst sc,
                                                          const uint8_t* __restrict__ mask, double alpha,
                                                          const double* __restrict__ xu, const double* __restrict__ xp,
                                                          const double* __restrict__ cu, const double* __restrict__ cp,
                                                          int nxc, const double* __restrict__ bu,
                                                          const double* __restrict__ bp, double omega, int remap,
                                                          double* __restrict__ yu, double* __restrict__ yp) {
  constexpr int W0 = TX + 2 * K, H0 = TY + 2 * K;
  __shared__ double su_[2][W0 * H0], sp_[2][W0 * H0];
  const int sx = nx + 1;
  const int ntx = (nx + TX) / TX;
  const int b = xcd_block(blockIdx.x, gridDim.x, remap);
  const int i0 = (b % ntx) * TX - K, j0 = (b / ntx) * TY - K;  // origin of the halo image
  const int tid = threadIdx.x;
  for (int p = tid; POST && p < W0 * H0; p += PGX_BLOCK) {
    const int gi = i0 + p % W0, gj = j0 + p / W0;
    const bool in = gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
    const int v = gj * sx + gi;
    if (POST) {
      double a = 0.0, c2 = 0.0;
      if (in) {
        a = xu[v];
        c2 = xp[v];
        if (cu) {
          const int sxc = nxc + 1;
          const int ic = gi >> 1, jc = gj >> 1;
          const int c0 = jc * sxc + ic, c1 = (jc + (gj & 1)) * sxc + (ic + (gi & 1));
          a += 0.5 * (cu[c0] + cu[c1]);
          c2 += 0.5 * (cp[c0] + cp[c1]);
        }
      }
      su_[0][p] = (in && mask[v]) ? 0.0 : a;  // pre-masked image
      sp_[0][p] = c2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 1; s <= K; ++s) {
    const int wr = W0 - 2 * s, hr = H0 - 2 * s;  // region of this sweep, offset s inside the image
    const double* srcu = su_[(s - 1) & 1];
    const double* srcp = sp_[(s - 1) & 1];
    double* dstu = su_[s & 1];
    double* dstp = sp_[s & 1];
    for (int p = tid; p < wr * hr; p += PGX_BLOCK) {
      const int li = s + p % wr, lj = s + p / wr;
      const int gi = i0 + li, gj = j0 + lj;
      const int q0 = lj * W0 + li;
      double ou = 0.0, op = 0.0;
      const bool in = gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
      if (in) {
        const int v = gj * sx + gi;
        StCoef c;
        st_load_coef(v, gi, gj, nx, ny, n, Kc, M, Dh, sc, mask, c);
        double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
        if (POST || s > 1) {
          st_rows_img(c, alpha, srcu, srcp, q0, W0, au, ap);
          xur = srcu[q0];
          xpr = srcp[q0];
        }
        st_jacobi(c, alpha, omega, au, ap, xur, xpr, bu[v], bp[v], ou, op);
        if (s == K) {
          yu[v] = ou;
          yp[v] = op;
        }
        if (c.rowbc) ou = 0.0;  // pre-masked image
      }
      if (s < K) {
        dstu[q0] = ou;
        dstp[q0] = op;
      }
    }
    if (s < K) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Row-mapped K-sweep smoother (the production kernel on uniform levels; k_st_smoothK above stays as the general-mesh /
// A-B reference, PGX_SMOOTH_ROWMAP=0).
// rocprofv3 counters on k_st_smoothK at 2048^2 (profiles/r02_smoother_pmc_before.json): the SIMDs issue instructions 71 % of
// the time and only 44 % of that is fp64 arithmetic - ~180 VALU instructions per point-sweep for ~40 flops: index
// arithmetic of the p -> (li, lj) loops, per-lane "is this link inside the grid" tests (each ld7h condition became a branch),
// coefficient selects, 64-bit address computations for 9 global loads.  LDS conflicts and LDS issue stalls are negligible.
// The kernel was instruction-bound, not bandwidth- or latency-bound, so this version removes instructions:
//  * a wave owns one 64-vertex ROW of the image at a time: the row index, every row base address and all bounds tests are
//    wave-uniform (scalar unit); per-lane work is arithmetic plus loads at "scalar base + lane";
//  * tiles whose whole image (tile + K halo) is interior take a FAST path with no bounds / Dirichlet tests at all: K and M
//    stencils are scalar constants (uniform grid), symmetric link pairs are summed before they are multiplied;
//  * (u, psi) are interleaved in LDS: one ds_read_b128 per neighbour instead of two ds_read_b64;
//  * boundary tiles (7 % at 2048^2) run the same general per-point code as k_st_smoothK.
// Image: 64 x (TY + 2K) vertices, tile = the inner (64 - 2K) x TY.  Same algebra as k_st_smoothK (different summation order).
// ------------------------------------------------------------------------------------------------
// The mirrored D links come from the neighbours, not from memory (round 3): D1[v - 1], D2[v - sx] and D3[v - sx - 1] are the stored
// values of the left / upper / upper-left neighbour - the first is the left lane's register (DPP wave shift), the other two sit in
// another wave's registers and cross LDS - so a vertex pulls 48 instead of 72 bytes of coefficients through the vector L1.  What
// the timing experiments behind this say: the launch is bound by the memory traffic of its coefficient
// streams at an effective 3-4 TB/s, not by the number of its load instructions.
__device__ __forceinline__ double lane_shr1(double x) {  // the value of lane - 1 (lane 0 keeps its own)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// interior tiles: no bounds / Dirichlet tests, scalar K and M stencils with symmetric link pairs pre-summed; the global loads
// of the NEXT row are in flight while the current row is computed (register double buffer)
template <int TY, int K, bool POST, typename DT>
__device__ __forceinline__ void st_smoothR_fast(int b, int nx, int n, int nfx, const DT* __restrict__ Dh,
                                                          const StConst& sc, double alpha, const double* __restrict__ xu,
                                                          const double* __restrict__ xp, const double* __restrict__ cu,
                                                          const double* __restrict__ cp, int nxc,
                                                          const double* __restrict__ bu, const double* __restrict__ bp,
                                                          double omega, double* __restrict__ yu, double* __restrict__ yp,
                                                          double2* img_a, double2* img_b, double2* exch) {
  // A wave owns the SAME image rows in every sweep: lj = wave + NW k.  Their iterate-independent data (7 D links, b_u, b_psi)
  // is loaded ONCE, all loads in flight together, and stays in registers for the K sweeps: a wave's chain of dependent
  // HBM round trips - which, not bandwidth or arithmetic, bounded the 4-wave version (6 + 14 sequential row iterations of
  // ~1-2 us per workgroup) - shrinks to one.
  constexpr int W = 64, TX = W - 2 * K, H0 = TY + 2 * K, NW = PGX_ROWMAP_BLOCK / 64, R = (H0 + NW - 1) / NW;
  const int sx = nx + 1;
  const int i0 = (1 + b % nfx) * TX - K, j0 = (1 + b / nfx) * TY - K;  // origin of the image
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  double2* const img0 = img_a;
  double2* const img1 = img_b;
  const double k0 = alpha * sc.K[0], k1 = alpha * 0.5 * (sc.K[1] + sc.K[2]), k3 = alpha * 0.5 * (sc.K[3] + sc.K[4]),
               k5 = alpha * 0.5 * (sc.K[5] + sc.K[6]);
  const double m0 = sc.M[0], m1 = 0.5 * (sc.M[1] + sc.M[2]), m3 = 0.5 * (sc.M[3] + sc.M[4]), m5 = 0.5 * (sc.M[5] + sc.M[6]);
  const double nb2 = -m0 * m0;
  const DT* const D1 = Dh + n;
  const DT* const D2 = Dh + 2 * (size_t)n;
  const DT* const D3 = Dh + 3 * (size_t)n;
  double rd[R][7], rbu[R], rbp[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < H0 - 1) {  // rows 1 .. H0-2 are updated; row 0 only hands its upward links to row 1
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      rd[k][3] = D2[v];
      rd[k][5] = D3[v];
      if (lj >= 1) {
        rd[k][0] = Dh[v];
        rd[k][1] = D1[v];
        rbu[k] = bu[v];
        rbp[k] = bp[v];
      }
    }
  }
  if (POST) {
    double xa[R], xc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < H0) {
        const int gj = j0 + lj;
        const unsigned v = (unsigned)(gj * sx + gi);
        xa[k] = xu[v];
        xc[k] = xp[v];
        if (cu) {
          const int sxc = nxc + 1;
          const int jc = gj >> 1, ic = gi >> 1;
          const unsigned c0 = (unsigned)(jc * sxc + ic), c1 = (unsigned)((jc + (gj & 1)) * sxc + ic + (gi & 1));
          xa[k] += 0.5 * (cu[c0] + cu[c1]);
          xc[k] += 0.5 * (cp[c0] + cp[c1]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < H0) img0[lj * W + lane] = make_double2(xa[k], xc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < H0 - 1) exch[lj * W + lane] = make_double2(rd[k][3], rd[k][5]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj >= 1 && lj < H0 - 1) {
      rd[k][2] = lane_shr1(rd[k][1]);                // D1[v - 1]
      rd[k][4] = exch[(lj - 1) * W + lane].x;        // D2[v - sx]
      rd[k][6] = exch[(lj - 1) * W + lane - 1].y;    // D3[v - sx - 1]; lane 0 (halo column, never updated) reads the guard band
    }
  }
#pragma unroll
  for (int s = 1; s <= K; ++s) {
    const double2* const src = ((s - 1) & 1) ? img1 : img0;
    double2* const dst = (s & 1) ? img1 : img0;
    const bool act = lane >= s && lane < W - s;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < s || lj >= H0 - s) continue;  // wave-uniform
      double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
      if (POST || s > 1) {
        const int q = lj * W + lane;
        const double2 x0 = src[q], x1 = src[q + 1], x2 = src[q - 1], x3 = src[q + W], x4 = src[q - W], x5 = src[q + W + 1],
                      x6 = src[q - W - 1];
        const double u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
        const double p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
        au = k0 * x0.x + k1 * u12 + k3 * u34 + k5 * u56 + m0 * x0.y + m1 * p12 + m3 * p34 + m5 * p56;
        ap = m0 * x0.x + m1 * u12 + m3 * u34 + m5 * u56 -
             (rd[k][0] * x0.y + rd[k][1] * x1.y + rd[k][2] * x2.y + rd[k][3] * x3.y + rd[k][4] * x4.y + rd[k][5] * x5.y +
              rd[k][6] * x6.y);
        xur = x0.x;
        xpr = x0.y;
      }
      const double su = rbu[k] - au, sp = rbp[k] - ap;
      const double det = fma(-k0, rd[k][0], nb2);  // < 0: k0 > 0, d0 >= 0, m0 > 0
      double rc = __builtin_amdgcn_rcp(det);
      rc = rc * (2.0 - det * rc);
      const double ou = xur + omega * ((-rd[k][0] * su - m0 * sp) * rc);
      const double op = xpr + omega * ((-m0 * su + k0 * sp) * rc);
      if (act) {
        if (s == K) {
          const unsigned v = (unsigned)((j0 + lj) * sx + gi);
          yu[v] = ou;
          yp[v] = op;
        } else {
          dst[lj * W + lane] = make_double2(ou, op);
        }
      }
    }
    if (s < K) __syncthreads();
  }
}

// boundary tiles (and every tile of a level without uniform stencils): the general per-point code of k_st_smoothK on the
// row mapping.  Tile index: row ty = 0 | rows 1..nfy: columns 0 and nfx+1.. | rows nfy+1..
// TB: boundary tiles are cut into TY/TB sub-tiles of TB rows, one workgroup each: a boundary workgroup is a chain of dependent
// loads (coefficient arrays, per-lane tests) of ~1 us per row iteration, and with TY + 6 rows per image that chain - 20-30 us -
// was the floor of EVERY level's launch time; TB = 4 gives 10-row images, 8 iterations instead of 21 (2.5x redundant work on
// 7 % of the tiles).
template <int TY, int TB, int K, bool POST>
__device__ __forceinline__ void st_smoothR_bnd(int b, int nx, int ny, int n, const RowmapGrid& g,
                                                         const double* __restrict__ Kc, const double* __restrict__ M,
                                                         const dsten_t* __restrict__ Dh, const StConst& sc,
                                                         const uint8_t* __restrict__ mask, double alpha,
                                                         const double* __restrict__ xu, const double* __restrict__ xp,
                                                         const double* __restrict__ cu, const double* __restrict__ cp,
                                                         int nxc, const double* __restrict__ bu,
                                                         const double* __restrict__ bp, double omega,
                                                         double* __restrict__ yu, double* __restrict__ yp, double2* img_a,
                                                         double2* img_b) {
  constexpr int W = 64, TX = W - 2 * K, H0 = TB + 2 * K, NSUB = TY / TB, NW = PGX_ROWMAP_BLOCK / 64;
  static_assert(TY % TB == 0, "sub-tiles must cover a tile");
  const int sx = nx + 1;
  int tx, ty;
  const int sub = b % NSUB;
  b /= NSUB;
  const int side = g.ntx - g.nfx;  // boundary tiles in a row that also holds fast tiles
  if (b < g.ntx) {
    tx = b;
    ty = 0;
  } else if ((b -= g.ntx) < g.nfy * side) {
    ty = 1 + b / side;
    const int r = b % side;
    tx = r == 0 ? 0 : g.nfx + r;
  } else {
    b -= g.nfy * side;
    ty = g.nfy + 1 + b / g.ntx;
    tx = b % g.ntx;
  }
  const int i0 = tx * TX - K, j0 = ty * TY + sub * TB - K;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  double2* const img0 = img_a;
  double2* const img1 = img_b;
  if (POST) {
    for (int lj = wave; lj < H0; lj += NW) {
      const int gj = j0 + lj;
      const bool in = gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
      double a = 0.0, c2 = 0.0;
      if (in) {
        const int v = gj * sx + gi;
        a = xu[v];
        c2 = xp[v];
        if (cu) {
          const int sxc = nxc + 1;
          const int ic = gi >> 1, jc = gj >> 1;
          const int c0 = jc * sxc + ic, c1 = (jc + (gj & 1)) * sxc + (ic + (gi & 1));
          a += 0.5 * (cu[c0] + cu[c1]);
          c2 += 0.5 * (cp[c0] + cp[c1]);
        }
        if (mask[v]) a = 0.0;  // pre-masked image: Dirichlet entries of u read as 0 by neighbours
      }
      img0[lj * W + lane] = make_double2(a, c2);
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 1; s <= K; ++s) {
    const double2* const src = ((s - 1) & 1) ? img1 : img0;
    double2* const dst = (s & 1) ? img1 : img0;
    const bool act = lane >= s && lane < W - s;
    for (int lj = s + wave; lj < H0 - s; lj += NW) {
      const int gj = j0 + lj;
      const int q = lj * W + lane;
      double ou = 0.0, op = 0.0;
      if (act && gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
        const int v = gj * sx + gi;
        StCoef c;
        st_load_coef(v, gi, gj, nx, ny, n, Kc, M, Dh, sc, mask, c);
        double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
        if (POST || s > 1) {
          const int off[7] = {0, 1, -1, W, -W, W + 1, -W - 1};
#pragma unroll
          for (int t = 0; t < 7; ++t) {  // out-of-grid / Dirichlet entries of the image are 0, links leaving the grid have zero coefficients
            const double2 xn = src[q + off[t]];
            au += alpha * c.kv[t] * xn.x + c.mv[t] * xn.y;
            ap += c.mv[t] * xn.x - c.dv[t] * xn.y;
          }
          const double2 x0 = src[q];
          xur = x0.x;
          xpr = x0.y;
        }
        st_jacobi(c, alpha, omega, au, ap, xur, xpr, bu[v], bp[v], ou, op);
        if (s == K) {
          yu[v] = ou;
          yp[v] = op;
        }
        if (c.rowbc) ou = 0.0;  // pre-masked image
      }
      if (s < K && act) dst[q] = make_double2(ou, op);
    }
    if (s < K) __syncthreads();
  }
}

// ONE launch per smoother call: blocks [0, nbnd) are the boundary tiles - they start first, so their long dependent-load
// chains overlap with the interior tiles that follow - blocks [nbnd, nbnd + nfast) the interior tiles.
template <int TY, int K, bool POST, bool D32 = false>
__global__ void __launch_bounds__(PGX_ROWMAP_BLOCK) k_st_smoothR(int nx, int ny, int n, RowmapGrid g, int nbnd,
                                                          const double* __restrict__ Kc, const double* __restrict__ M,
                                                          const dsten_t* __restrict__ Dh, const float* __restrict__ Dh32, StConst sc,
                                                          const uint8_t* __restrict__ mask, double alpha,
                                                          const double* __restrict__ xu, const double* __restrict__ xp,
                                                          const double* __restrict__ cu, const double* __restrict__ cp,
                                                          int nxc, const double* __restrict__ bu,
                                                          const double* __restrict__ bp, double omega, int remap,
                                                          double* __restrict__ yu, double* __restrict__ yp) {
  constexpr int W = 64, H0 = TY + 2 * K, PAD = W + 1;
  __shared__ double2 img_[3][H0 * W + 2 * PAD];  // guard band: inactive edge lanes read (and discard) one entry outside a row;
                                                  // [2]: the (D2, D3) links every image row hands to the row below it
  const int blk = blockIdx.x;
  if (blk < nbnd)
    st_smoothR_bnd<TY, 4, K, POST>(blk, nx, ny, n, g, Kc, M, Dh, sc, mask, alpha, xu, xp, cu, cp, nxc, bu, bp, omega, yu, yp,
                                img_[0] + PAD, img_[1] + PAD);
  else if (D32)  // interior tiles read the single-precision copy of D (finest level)
    st_smoothR_fast<TY, K, POST, float>(xcd_block(blk - nbnd, gridDim.x - nbnd, remap), nx, n, g.nfx, Dh32, sc, alpha, xu, xp, cu, cp,
                                        nxc, bu, bp, omega, yu, yp, img_[0] + PAD, img_[1] + PAD, img_[2] + PAD);
  else
    st_smoothR_fast<TY, K, POST, dsten_t>(xcd_block(blk - nbnd, gridDim.x - nbnd, remap), nx, n, g.nfx, Dh, sc, alpha, xu, xp, cu, cp,
                                          nxc, bu, bp, omega, yu, yp, img_[0] + PAD, img_[1] + PAD, img_[2] + PAD);
}

template <int TYR, int KS = 3>
static void launch_rowmap(hipStream_t st, int post, const GridLevel& L, const StConst& sc, double alpha, const double* xu,
                          const double* xp, const GridLevel* C, const double* cu, const double* cp, const double* bu,
                          const double* bp, double omega, int remap, double* yu, double* yp) {
  const RowmapGrid g = rowmap_grid<TYR, KS>(L.nx, L.ny, L.interior_free);
  const int nfast = g.nfx * g.nfy, nbnd = (g.ntx * g.nty - nfast) * (TYR / 4);  // boundary tiles: sub-tiles of 4 rows
  dim3 grid(nbnd + nfast), block(PGX_ROWMAP_BLOCK);
  if (L.Dh32 && KS == 3) {
    if (post)
      hipLaunchKernelGGL((k_st_smoothR<TYR, 3, true, true>), grid, block, 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, L.Dh32, sc,
                         L.mask, alpha, xu, xp, cu, cp, C ? C->nx : 0, bu, bp, omega, remap, yu, yp);
    else
      hipLaunchKernelGGL((k_st_smoothR<TYR, 3, false, true>), grid, block, 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, L.Dh32, sc,
                         L.mask, alpha, nullptr, nullptr, nullptr, nullptr, 0, bu, bp, omega, remap, yu, yp);
    return;
  }
  if (post)
    hipLaunchKernelGGL((k_st_smoothR<TYR, KS, true>), grid, block, 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, nullptr, sc, L.mask,
                       alpha, xu, xp, cu, cp, C ? C->nx : 0, bu, bp, omega, remap, yu, yp);
  else
    hipLaunchKernelGGL((k_st_smoothR<TYR, KS, false>), grid, block, 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, nullptr, sc, L.mask,
                       alpha, nullptr, nullptr, nullptr, nullptr, 0, bu, bp, omega, remap, yu, yp);
}

// K = 6 is available on levels with uniform interior stencils (row-mapped kernels only)
int pgxk_st_smooth6_ok(const GridLevel& L) {
  static PgxTuneInt t_rowmap("PGX_SMOOTH_ROWMAP", 1);
  const int rowmap = t_rowmap.get();
  return rowmap && L.uniform;
}

// post=0: S^K(0);  post=1: S^K((xu,xp) + P (cu,cp)) (cu may be null)   -- out of place; K in {2,3,6}
void pgxk_st_smoothK(hipStream_t st, int K, int post, const GridLevel& L, double alpha, const double* xu,
                     const double* xp, const GridLevel* C, const double* cu, const double* cp, const double* bu,
                     const double* bp, double omega, int remap, double* yu, double* yp) {
  if (K == 2) {
    pgxk_st_smooth2(st, post, L, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    return;
  }
  const StConst sc = make_stconst(L);
  static PgxTuneInt t_rowmap("PGX_SMOOTH_ROWMAP", 1);
  const int rowmap = t_rowmap.get();
  if (K == 6) {  // small levels (pgxk_st_smooth6_ok): SIX sweeps per launch - one latency-bound launch instead of two
    static PgxTuneInt t_ty6("PGX_K6_TY", 0);
    const int ty6 = t_ty6.get();
    if (ty6 == 16)  // measured at 2049^2: 16-row tiles 323 ms per solve, 8-row tiles 298 ms, three-sweep launches (default) 304 ms
      launch_rowmap<16, 6>(st, post, L, sc, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    else
      launch_rowmap<8, 6>(st, post, L, sc, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    return;
  }
  if (rowmap && L.uniform) {  // row-mapped kernels: image 64 x (TY + 6), tile 58 x TY; interior tiles + boundary tiles
    // tile height by level size (measured, us per launch at 2049^2 / 1025^2 / 513^2 / 257^2 vertices; PGX_ROWMAP_TY forces one)
    static PgxTuneInt t_ty("PGX_ROWMAP_TY", 0);
    const int ty_env = t_ty.get();
    const int ty = ty_env ? ty_env : (L.n >= 2000000 ? 16 : L.n >= 500000 ? 8 : 4);
    if (ty == 4)
      launch_rowmap<4>(st, post, L, sc, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    else if (ty == 8)
      launch_rowmap<8>(st, post, L, sc, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    else
      launch_rowmap<16>(st, post, L, sc, alpha, xu, xp, C, cu, cp, bu, bp, omega, remap, yu, yp);
    return;
  }
  constexpr int TX = PGX_TILE_X, TY = PGX_TILE_Y;
  const int ntx = (L.nx + TX) / TX, nty = (L.ny + TY) / TY;
  dim3 grid(ntx * nty), block(PGX_BLOCK);
  if (post)
    hipLaunchKernelGGL((k_st_smoothK<TX, TY, 3, true>), grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask,
                       alpha, xu, xp, cu, cp, C ? C->nx : 0, bu, bp, omega, remap, yu, yp);
  else
    hipLaunchKernelGGL((k_st_smoothK<TX, TY, 3, false>), grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc,
                       L.mask, alpha, nullptr, nullptr, nullptr, nullptr, 0, bu, bp, omega, remap, yu, yp);
}

// b_c = P^T (b - J x) without a residual round trip through HBM.  (A first version with one thread per COARSE vertex
// evaluating its 7 fine residuals itself measured 189 us on level 0 - strided gathers - against 121 us for separate
// residual + restriction launches; the tile version below takes 65 us.)
// A block owns a CXxCY tile of COARSE vertices; the fine residual is
// evaluated ONCE per fine vertex of the (2CX+1)x(2CY+1) footprint with row-contiguous (coalesced) accesses
// into LDS, then each thread restricts one coarse vertex from LDS.
template <int CX, int CY>
__global__ void __launch_bounds__(PGX_BLOCK) k_st_resid_restrict_t(int nx, int ny, int n, const double* __restrict__ K,
                                                                   const double* __restrict__ M,
                                                                   const dsten_t* __restrict__ Dh, StConst sc,
                                                                   const uint8_t* __restrict__ mask, double alpha,
                                                                   const double* __restrict__ xu,
                                                                   const double* __restrict__ xp,
                                                                   const double* __restrict__ bu,
                                                                   const double* __restrict__ bp, int nxc, int nyc,
                                                                   const uint8_t* __restrict__ mask_c, int remap,
                                                                   double* __restrict__ cbu, double* __restrict__ cbp) {
  static_assert(CX * CY == PGX_BLOCK, "one coarse vertex per thread");
  constexpr int W = 2 * CX + 1, H = 2 * CY + 1;
  __shared__ double sru[W * H], srp[W * H];
  const int sx = nx + 1, sxc = nxc + 1;
  const int ntx = (nxc + CX) / CX;
  const int b = xcd_block(blockIdx.x, gridDim.x, remap);
  const int I0 = (b % ntx) * CX, J0 = (b / ntx) * CY;
  const int tid = threadIdx.x;
  const int off[7] = {0, 1, -1, sx, -sx, sx + 1, -sx - 1};
  for (int p = tid; p < W * H; p += PGX_BLOCK) {
    const int gi = 2 * I0 - 1 + p % W, gj = 2 * J0 - 1 + p / W;
    double ru = 0.0, rp = 0.0;
    if (gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
      const int v = gj * sx + gi;
      StCoef c;
      st_load_coef(v, gi, gj, nx, ny, n, K, M, Dh, sc, mask, c);
      double xun[7], xpn[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const int nb = c.ok[s] ? v + off[s] : v;
        xun[s] = mask[nb] ? 0.0 : xu[nb];
        xpn[s] = xp[nb];
      }
      double au, ap;
      st_rows(c, alpha, xun, xpn, au, ap);
      if (c.rowbc) au = xu[v];
      ru = bu[v] - au;
      rp = bp[v] - ap;
    }
    sru[p] = ru;
    srp[p] = rp;
  }
  __syncthreads();
  const int I = I0 + tid % CX, J = J0 + tid / CX;
  if (I > nxc || J > nyc) return;
  constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
  constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
  const int li = 2 * (tid % CX) + 1, lj = 2 * (tid / CX) + 1;  // position of fine vertex (2I,2J) in the LDS image
  double su = 0.0, sp = 0.0;
#pragma unroll
  for (int o = 0; o < 7; ++o) {  // out-of-grid fine vertices hold 0 in the image
    const int q = (lj + OY[o]) * W + (li + OX[o]);
    const double w = pw(OX[o], OY[o]);
    su += w * sru[q];
    sp += w * srp[q];
  }
  const int C = J * sxc + I;
  cbu[C] = mask_c[C] ? 0.0 : su;
  cbp[C] = sp;
}

// ------------------------------------------------------------------------------------------------
// Row-mapped b_c = P^T (b - J x): the same treatment as k_st_smoothR (the tile kernel above was instruction-bound in the same
// way).  A workgroup of 8 waves owns CX x CY = 30 x 8 coarse vertices: (1) the iterate on the 63 x 19 fine footprint + 1 halo
// goes into an LDS image, (u, psi) interleaved, a wave per row; (2) a wave per fine row evaluates the residual on 61 x 17
// vertices into a second image - interior tiles with scalar K / M stencils and no tests; (3) wave w restricts coarse row w.
// Boundary tiles (blocks [0, nbnd), scheduled first) run the general per-point code on the same mapping.
// ------------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void st_rr_tile(int tx, int ty, int nx, int ny, int n, const double* __restrict__ K,
                                           const double* __restrict__ M, const dsten_t* __restrict__ Dh, const StConst& sc,
                                           const uint8_t* __restrict__ mask, double alpha, const double* __restrict__ xu,
                                           const double* __restrict__ xp, const double* __restrict__ bu,
                                           const double* __restrict__ bp, int nxc, int nyc, const uint8_t* __restrict__ mask_c,
                                           double* __restrict__ cbu, double* __restrict__ cbp, double2* ximg, double2* rimg) {
  constexpr int W = 64, CX = 30, CY = 8, HX = 2 * CY + 3, HR = 2 * CY + 1, NW = PGX_ROWMAP_BLOCK / 64;
  const int sx = nx + 1, sxc = nxc + 1;
  const int I0 = tx * CX, J0 = ty * CY;
  const int i0 = 2 * I0 - 2, j0 = 2 * J0 - 2;  // origin of the x image; the residual image starts one row / column further in
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  // (1) iterate image
  for (int lj = wave; lj < HX; lj += NW) {
    const int gj = j0 + lj;
    double a = 0.0, c2 = 0.0;
    if (FAST) {
      const unsigned v = (unsigned)(gj * sx + gi);
      a = xu[v];
      c2 = xp[v];
    } else if (gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
      const int v = gj * sx + gi;
      a = mask[v] ? 0.0 : xu[v];  // pre-masked image
      c2 = xp[v];
    }
    ximg[lj * W + lane] = make_double2(a, c2);
  }
  __syncthreads();
  // (2) fine residual on image rows 1 .. HX-2, columns 1 .. 62
  const bool act = lane >= 1 && lane < W - 1;
  if (FAST) {
    const double k0 = alpha * sc.K[0], k1 = alpha * 0.5 * (sc.K[1] + sc.K[2]), k3 = alpha * 0.5 * (sc.K[3] + sc.K[4]),
                 k5 = alpha * 0.5 * (sc.K[5] + sc.K[6]);
    const double m0 = sc.M[0], m1 = 0.5 * (sc.M[1] + sc.M[2]), m3 = 0.5 * (sc.M[3] + sc.M[4]), m5 = 0.5 * (sc.M[5] + sc.M[6]);
    const dsten_t* const D1 = Dh + n;
    const dsten_t* const D2 = Dh + 2 * (size_t)n;
    const dsten_t* const D3 = Dh + 3 * (size_t)n;
#pragma unroll
    for (int k = 0; k < (HR + NW - 1) / NW; ++k) {
      const int lj = 1 + wave + NW * k;
      if (lj > HR) continue;  // wave-uniform
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      const double d0 = Dh[v], d1 = D1[v], d2 = D1[v - 1], d3 = D2[v], d4 = D2[v - sx], d5 = D3[v], d6 = D3[v - sx - 1];
      const double buv = bu[v], bpv = bp[v];
      const int q = lj * W + lane;
      const double2 x0 = ximg[q], x1 = ximg[q + 1], x2 = ximg[q - 1], x3 = ximg[q + W], x4 = ximg[q - W], x5 = ximg[q + W + 1],
                    x6 = ximg[q - W - 1];
      const double u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
      const double p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
      const double au = k0 * x0.x + k1 * u12 + k3 * u34 + k5 * u56 + m0 * x0.y + m1 * p12 + m3 * p34 + m5 * p56;
      const double ap = m0 * x0.x + m1 * u12 + m3 * u34 + m5 * u56 -
                        (d0 * x0.y + d1 * x1.y + d2 * x2.y + d3 * x3.y + d4 * x4.y + d5 * x5.y + d6 * x6.y);
      if (act) rimg[(lj - 1) * W + lane] = make_double2(buv - au, bpv - ap);
    }
  } else {
    for (int lj = 1 + wave; lj <= HR; lj += NW) {
      const int gj = j0 + lj;
      double ru = 0.0, rp = 0.0;
      if (act && gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
        const int v = gj * sx + gi;
        StCoef c;
        st_load_coef(v, gi, gj, nx, ny, n, K, M, Dh, sc, mask, c);
        const int q = lj * W + lane;
        const int off[7] = {0, 1, -1, W, -W, W + 1, -W - 1};
        double au = 0.0, ap = 0.0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {  // out-of-grid / Dirichlet entries of the image are 0; links leaving the grid have zero coefficients
          const double2 xn = ximg[q + off[t]];
          au += alpha * c.kv[t] * xn.x + c.mv[t] * xn.y;
          ap += c.mv[t] * xn.x - c.dv[t] * xn.y;
        }
        if (c.rowbc) au = xu[v];
        ru = bu[v] - au;
        rp = bp[v] - ap;
      }
      if (act) rimg[(lj - 1) * W + lane] = make_double2(ru, rp);  // out-of-grid fine vertices hold 0
    }
  }
  __syncthreads();
  // (3) restriction: wave w -> coarse row J0 + w, lane -> coarse column I0 + lane.  Fine vertex (2I, 2J) sits at residual-image
  // row 2w + 1 (image rows start at fine row 2 J0 - 1) and column 2 lane + 2 (lane index of fine column 2 I0 + 2 lane)
  if (wave < CY && lane < CX) {
    const int I = I0 + lane, J = J0 + wave;
    if (FAST || (I <= nxc && J <= nyc)) {
      const int q = (2 * wave + 1) * W + 2 * lane + 2;
      const double2 r0 = rimg[q], r1 = rimg[q + 1], r2 = rimg[q - 1], r3 = rimg[q + W], r4 = rimg[q - W], r5 = rimg[q + W + 1],
                    r6 = rimg[q - W - 1];
      const double su = r0.x + 0.5 * (r1.x + r2.x + r3.x + r4.x + r5.x + r6.x);
      const double sp = r0.y + 0.5 * (r1.y + r2.y + r3.y + r4.y + r5.y + r6.y);
      const int C = J * sxc + I;
      cbu[C] = (!FAST && mask_c[C]) ? 0.0 : su;
      cbp[C] = sp;
    }
  }
}

__global__ void __launch_bounds__(PGX_ROWMAP_BLOCK) k_st_resid_restrict_r(int nx, int ny, int n, RrGrid g, int nbnd,
                                                                          const double* __restrict__ K,
                                                                          const double* __restrict__ M,
                                                                          const dsten_t* __restrict__ Dh, StConst sc,
                                                                          const uint8_t* __restrict__ mask, double alpha,
                                                                          const double* __restrict__ xu,
                                                                          const double* __restrict__ xp,
                                                                          const double* __restrict__ bu,
                                                                          const double* __restrict__ bp, int nxc, int nyc,
                                                                          const uint8_t* __restrict__ mask_c, int remap,
                                                                          double* __restrict__ cbu, double* __restrict__ cbp) {
  constexpr int W = 64, CY = 8, HX = 2 * CY + 3, HR = 2 * CY + 1, PAD = W + 1;
  __shared__ double2 ximg_[HX * W + 2 * PAD], rimg_[HR * W + 2 * PAD];
  int b = blockIdx.x;
  if (b < nbnd) {
    int tx, ty;
    const int side = g.ntx - g.nfx;
    if (b < g.ntx) {
      tx = b;
      ty = 0;
    } else if ((b -= g.ntx) < g.nfy * side) {
      ty = 1 + b / side;
      const int r = b % side;
      tx = r == 0 ? 0 : g.nfx + r;
    } else {
      b -= g.nfy * side;
      ty = g.nfy + 1 + b / g.ntx;
      tx = b % g.ntx;
    }
    st_rr_tile<false>(tx, ty, nx, ny, n, K, M, Dh, sc, mask, alpha, xu, xp, bu, bp, nxc, nyc, mask_c, cbu, cbp, ximg_ + PAD,
                      rimg_ + PAD);
  } else {
    b = xcd_block(b - nbnd, gridDim.x - nbnd, remap);
    st_rr_tile<true>(1 + b % g.nfx, 1 + b / g.nfx, nx, ny, n, K, M, Dh, sc, mask, alpha, xu, xp, bu, bp, nxc, nyc, mask_c, cbu, cbp,
                     ximg_ + PAD, rimg_ + PAD);
  }
}

void pgxk_st_resid_restrict(hipStream_t st, const GridLevel& L, double alpha, const double* xu, const double* xp,
                            const double* bu, const double* bp, const GridLevel& C, int remap, double* cbu,
                            double* cbp) {
  static PgxTuneInt t_rowmap("PGX_SMOOTH_ROWMAP", 1);
  const int rowmap = t_rowmap.get();
  if (rowmap && L.uniform) {
    constexpr int CX = 30, CY = 8;
    RrGrid g;
    g.ntx = (C.nx + CX) / CX;
    g.nty = (C.ny + CY) / CY;
    // interior <=> every vertex of the x image is strictly inside the fine grid: 2 tx CX - 2 >= 1, 2 tx CX - 2 + 63 <= nx - 1,
    // 2 ty CY - 2 >= 1, 2 ty CY - 2 + (2 CY + 2) <= ny - 1
    g.nfx = (L.nx - 62) >= 2 * CX ? (L.nx - 62) / (2 * CX) : 0;
    g.nfy = (L.ny - 1 - 2 * CY) >= 2 * CY ? (L.ny - 1 - 2 * CY) / (2 * CY) : 0;
    g.nfx = std::min(g.nfx, g.ntx - 1);
    g.nfy = std::min(g.nfy, g.nty - 1);
    if (!L.interior_free || !C.interior_free || g.nfx <= 0 || g.nfy <= 0) g.nfx = g.nfy = 0;
    const int nfast = g.nfx * g.nfy, nbnd = g.ntx * g.nty - nfast;
    hipLaunchKernelGGL(k_st_resid_restrict_r, dim3(nbnd + nfast), dim3(PGX_ROWMAP_BLOCK), 0, st, L.nx, L.ny, L.n, g, nbnd, L.K,
                       L.M, L.Dh, make_stconst(L), L.mask, alpha, xu, xp, bu, bp, C.nx, C.ny, C.mask, remap, cbu, cbp);
    return;
  }
  constexpr int CX = 32, CY = 8;
  const int ntx = (C.nx + CX) / CX, nty = (C.ny + CY) / CY;
  hipLaunchKernelGGL((k_st_resid_restrict_t<CX, CY>), dim3(ntx * nty), dim3(PGX_BLOCK), 0, st, L.nx, L.ny, L.n, L.K,
                     L.M, L.Dh, make_stconst(L), L.mask, alpha, xu, xp, bu, bp, C.nx, C.ny, C.mask, remap, cbu, cbp);
}

// ------------------------------------------------------------------------------------------------
// Row-mapped matrix-free operator apply  y = J x  on a structured level: the outer-Krylov "SpMV" of structured P1 handles
// (pgx_api.hip: spmv_dev).  J = [[aK, M], [M, -D(psi)]] is never formed as a CSR matrix here: K and M are the seven
// constants of the uniform mesh, D(psi) its half-stored stencil (centre + three forward links, 32 B per vertex; the backward
// links are the neighbours' forward links), so one apply moves x (16 B) + y (16 B) + D (32 B) + mask (1 B) = 65 B per
// vertex instead of the 232 B per row of the block-CSR stream (k_bspmv_stream: 7 x 28 B + row pointer + x + y).
// Mapping as in k_st_resid_restrict_r: a workgroup of 8 waves owns 62 x RY vertices; (1) a wave per row stages
// (u, psi) interleaved on the (RY + 2) x 64 footprint in LDS; (2) a wave per row evaluates both rows of J from the image
// - interior tiles with the scalar stencils, pair sums and no tests, boundary tiles (blocks [0, nbnd), scheduled first)
// through the general per-point code with Dirichlet rows / columns as identity.
// ------------------------------------------------------------------------------------------------
#ifndef PGX_SPMV_RY
#define PGX_SPMV_RY 27  // (round 5: 24 -> 27 image rows per tile: 2508 instead of 2838 workgroups at 2048^2, 49.6-49.9 against 50.4-50.7 us on the same box)
#endif
// XF: the iterate comes as ONE interleaved (u, psi) float2 field - what the single-precision V-cycle leaves (round 5: the FGMRES Z_j
// are outputs of that cycle; stored and re-read as fp64 they were twice the bytes for no information) - xu then points at it
template <bool XF>
__device__ __forceinline__ double2 st_load_x(const double* __restrict__ xu, const double* __restrict__ xp, unsigned v) {
  if (XF) {
    const float2 t = reinterpret_cast<const float2*>(xu)[v];
    return make_double2((double)t.x, (double)t.y);
  }
  return make_double2(xu[v], xp[v]);
}
template <bool FAST, bool XF = false>
__device__ __forceinline__ void st_spmv_tile(int tx, int ty, int nx, int ny, int n, const double* __restrict__ K,
                                             const double* __restrict__ M, const dsten_t* __restrict__ Dh, const StConst& sc,
                                             const uint8_t* __restrict__ mask, double alpha, const double* __restrict__ xu,
                                             const double* __restrict__ xp, double* __restrict__ yu, double* __restrict__ yp,
                                             double2* ximg) {
  constexpr int W = 64, CXS = 62, RY = PGX_SPMV_RY, HX = RY + 2, NW = PGX_ROWMAP_BLOCK / 64;
  const int sx = nx + 1;
  const int i0 = tx * CXS - 1, j0 = ty * RY - 1;  // image origin: one halo column / row before the tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  for (int lj = wave; lj < HX; lj += NW) {
    const int gj = j0 + lj;
    double a = 0.0, c2 = 0.0;
    if (FAST) {
      const unsigned v = (unsigned)(gj * sx + gi);
      const double2 t = st_load_x<XF>(xu, xp, v);
      a = t.x;
      c2 = t.y;
    } else if (gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
      const int v = gj * sx + gi;
      const double2 t = st_load_x<XF>(xu, xp, (unsigned)v);
      a = mask[v] ? 0.0 : t.x;  // pre-masked image: Dirichlet columns of the u block contribute nothing
      c2 = t.y;
    }
    ximg[lj * W + lane] = make_double2(a, c2);
  }
  __syncthreads();
  const bool act = lane >= 1 && lane < W - 1;
  if (FAST) {
    const double k0 = alpha * sc.K[0], k1 = alpha * 0.5 * (sc.K[1] + sc.K[2]), k3 = alpha * 0.5 * (sc.K[3] + sc.K[4]),
                 k5 = alpha * 0.5 * (sc.K[5] + sc.K[6]);
    const double m0 = sc.M[0], m1 = 0.5 * (sc.M[1] + sc.M[2]), m3 = 0.5 * (sc.M[3] + sc.M[4]), m5 = 0.5 * (sc.M[5] + sc.M[6]);
    const dsten_t* const D1 = Dh + n;
    const dsten_t* const D2 = Dh + 2 * (size_t)n;
    const dsten_t* const D3 = Dh + 3 * (size_t)n;
#pragma unroll
    for (int k = 0; k < (RY + NW - 1) / NW; ++k) {
      const int lj = 1 + wave + NW * k;
      if (lj > RY) continue;  // wave-uniform
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      const double d0 = Dh[v], d1 = D1[v], d2 = D1[v - 1], d3 = D2[v], d4 = D2[v - sx], d5 = D3[v], d6 = D3[v - sx - 1];
      const int q = lj * W + lane;
      const double2 x0 = ximg[q], x1 = ximg[q + 1], x2 = ximg[q - 1], x3 = ximg[q + W], x4 = ximg[q - W], x5 = ximg[q + W + 1],
                    x6 = ximg[q - W - 1];
      const double u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
      const double p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
      const double au = k0 * x0.x + k1 * u12 + k3 * u34 + k5 * u56 + m0 * x0.y + m1 * p12 + m3 * p34 + m5 * p56;
      const double ap = m0 * x0.x + m1 * u12 + m3 * u34 + m5 * u56 -
                        (d0 * x0.y + d1 * x1.y + d2 * x2.y + d3 * x3.y + d4 * x4.y + d5 * x5.y + d6 * x6.y);
      if (act) {
        __builtin_nontemporal_store(au, yu + v);
        __builtin_nontemporal_store(ap, yp + v);
      }
    }
  } else {
    for (int lj = 1 + wave; lj <= RY; lj += NW) {
      const int gj = j0 + lj;
      if (act && gi >= 0 && gi <= nx && gj >= 0 && gj <= ny) {
        const int v = gj * sx + gi;
        StCoef c;
        st_load_coef(v, gi, gj, nx, ny, n, K, M, Dh, sc, mask, c);
        const int q = lj * W + lane;
        const int off[7] = {0, 1, -1, W, -W, W + 1, -W - 1};
        double au = 0.0, ap = 0.0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {  // out-of-grid / Dirichlet entries of the image are 0; links leaving the grid have zero coefficients
          const double2 xn = ximg[q + off[t]];
          au += alpha * c.kv[t] * xn.x + c.mv[t] * xn.y;
          ap += c.mv[t] * xn.x - c.dv[t] * xn.y;
        }
        yu[v] = c.rowbc ? st_load_x<XF>(xu, xp, (unsigned)v).x : au;
        yp[v] = ap;
      }
    }
  }
}

// Interior tiles, round 4: ONE batch of global loads per wave - the iterate on its image rows AND the four stored D links of the
// same rows, all in flight before the first LDS store - instead of the iterate first and, behind the barrier, seven D loads per
// vertex (three of them re-reads of the neighbours' links).  The mirrored links come from the neighbours as in the smoother: the
// left lane's register (DPP) and the (D2, D3) pair every row hands to the row above it through LDS.
template <bool D4, bool XF = false>
__device__ __forceinline__ void st_spmv_fast(int tx, int ty, int nx, int n, const dsten_t* __restrict__ Dh, const double4* __restrict__ Dd4,
                                             const StConst& sc, double alpha, const double* __restrict__ xu,
                                             const double* __restrict__ xp, double* __restrict__ yu, double* __restrict__ yp,
                                             double2* ximg, double2* exch) {
  constexpr int W = 64, CXS = 62, RY = PGX_SPMV_RY, HX = RY + 2, NW = PGX_ROWMAP_BLOCK / 64, R = (HX + NW - 1) / NW;
  const int sx = nx + 1;
  const int i0 = tx * CXS - 1, j0 = ty * RY - 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  const dsten_t* const D1 = Dh + n;
  const dsten_t* const D2 = Dh + 2 * (size_t)n;
  const dsten_t* const D3 = Dh + 3 * (size_t)n;
  double2 xa[R];
  double d0[R], d1[R], d3[R], d5[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < HX) {
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      xa[k] = st_load_x<XF>(xu, xp, v);
      if (lj <= RY) {  // row 0 only hands its upward links to row 1
        if (D4) {
          const double4 q = Dd4[v];
          d0[k] = q.x;
          d1[k] = q.y;
          d3[k] = q.z;
          d5[k] = q.w;
        } else {
          d3[k] = D2[v];
          d5[k] = D3[v];
          if (lj >= 1) {
            d0[k] = Dh[v];
            d1[k] = D1[v];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < HX) ximg[lj * W + lane] = xa[k];
    if (lj <= RY) exch[lj * W + lane] = make_double2(d3[k], d5[k]);
  }
  __syncthreads();
  const bool act = lane >= 1 && lane < W - 1;
  const double k0 = alpha * sc.K[0], k1 = alpha * 0.5 * (sc.K[1] + sc.K[2]), k3 = alpha * 0.5 * (sc.K[3] + sc.K[4]),
               k5 = alpha * 0.5 * (sc.K[5] + sc.K[6]);
  const double m0 = sc.M[0], m1 = 0.5 * (sc.M[1] + sc.M[2]), m3 = 0.5 * (sc.M[3] + sc.M[4]), m5 = 0.5 * (sc.M[5] + sc.M[6]);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < 1 || lj > RY) continue;  // wave-uniform
    const double d2 = lane_shr1(d1[k]), d4 = exch[(lj - 1) * W + lane].x, d6 = exch[(lj - 1) * W + lane - 1].y;
    const int q = lj * W + lane;
    const double2 x0 = ximg[q], x1 = ximg[q + 1], x2 = ximg[q - 1], x3 = ximg[q + W], x4 = ximg[q - W], x5 = ximg[q + W + 1],
                  x6 = ximg[q - W - 1];
    const double u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
    const double p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
    const double au = k0 * x0.x + k1 * u12 + k3 * u34 + k5 * u56 + m0 * x0.y + m1 * p12 + m3 * p34 + m5 * p56;
    const double ap = m0 * x0.x + m1 * u12 + m3 * u34 + m5 * u56 -
                      (d0[k] * x0.y + d1[k] * x1.y + d2 * x2.y + d3[k] * x3.y + d4 * x4.y + d5[k] * x5.y + d6 * x6.y);
    if (act) {
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      __builtin_nontemporal_store(au, yu + v);
      __builtin_nontemporal_store(ap, yp + v);
    }
  }
}

template <bool XF>
__global__ void __launch_bounds__(PGX_ROWMAP_BLOCK) k_st_spmv_r(int nx, int ny, int n, RrGrid g, int nbnd,
                                                                const double* __restrict__ K, const double* __restrict__ M,
                                                                const dsten_t* __restrict__ Dh, const double4* __restrict__ Dd4,
                                                                StConst sc, const uint8_t* __restrict__ mask, double alpha,
                                                                const double* __restrict__ xu, const double* __restrict__ xp,
                                                                int remap, double* __restrict__ yu, double* __restrict__ yp) {
  constexpr int W = 64, HX = PGX_SPMV_RY + 2, PAD = W + 1;
  __shared__ double2 ximg_[HX * W + 2 * PAD], exch_[HX * W + 2 * PAD];
  int b = blockIdx.x;
  if (b < nbnd) {  // same enumeration of the boundary frame as k_st_resid_restrict_r
    int tx, ty;
    const int side = g.ntx - g.nfx;
    if (b < g.ntx) {
      tx = b;
      ty = 0;
    } else if ((b -= g.ntx) < g.nfy * side) {
      ty = 1 + b / side;
      const int r = b % side;
      tx = r == 0 ? 0 : g.nfx + r;
    } else {
      b -= g.nfy * side;
      ty = g.nfy + 1 + b / g.ntx;
      tx = b % g.ntx;
    }
    st_spmv_tile<false, XF>(tx, ty, nx, ny, n, K, M, Dh, sc, mask, alpha, xu, xp, yu, yp, ximg_ + PAD);
  } else {
    b = xcd_block(b - nbnd, gridDim.x - nbnd, remap);
    if (Dd4)
      st_spmv_fast<true, XF>(1 + b % g.nfx, 1 + b / g.nfx, nx, n, Dh, Dd4, sc, alpha, xu, xp, yu, yp, ximg_ + PAD, exch_ + PAD);
    else
      st_spmv_fast<false, XF>(1 + b % g.nfx, 1 + b / g.nfx, nx, n, Dh, Dd4, sc, alpha, xu, xp, yu, yp, ximg_ + PAD, exch_ + PAD);
  }
}

// y = J x on a structured level, matrix-free (see above); levels without uniform interior stencils take k_st_apply<0>
void pgxk_st_spmv(hipStream_t st, const GridLevel& L, double alpha, const double* xu, const double* xp, int remap, double* yu,
                  double* yp, const float2* xf) {
  if (!L.uniform) {
    pgxk_st_apply(st, 0, L, alpha, xu, xp, nullptr, nullptr, 0.0, remap ? 2 : 0, yu, yp);
    return;
  }
  constexpr int CXS = 62, RY = PGX_SPMV_RY;
  RrGrid g;
  g.ntx = (L.nx + CXS) / CXS;  // ceil((nx + 1) / CXS)
  g.nty = (L.ny + RY) / RY;
  // interior <=> every vertex of the image is strictly inside the grid: tx CXS - 1 >= 1, tx CXS + 62 <= nx - 1; ty RY - 1 >= 1,
  // ty RY + RY <= ny - 1
  g.nfx = (L.nx - 1 - 62) >= CXS ? (L.nx - 1 - 62) / CXS : 0;
  g.nfy = (L.ny - 1 - RY) >= RY ? (L.ny - 1 - RY) / RY : 0;
  g.nfx = std::min(g.nfx, g.ntx - 1);
  g.nfy = std::min(g.nfy, g.nty - 1);
  if (!L.interior_free || g.nfx <= 0 || g.nfy <= 0) g.nfx = g.nfy = 0;
  const int nfast = g.nfx * g.nfy, nbnd = g.ntx * g.nty - nfast;
  if (xf)  // the iterate as one float2 field (pgxk_st_spmv_f2_ok levels only)
    hipLaunchKernelGGL(k_st_spmv_r<true>, dim3(nbnd + nfast), dim3(PGX_ROWMAP_BLOCK), 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, L.Dd4,
                       make_stconst(L), L.mask, alpha, reinterpret_cast<const double*>(xf), nullptr, remap, yu, yp);
  else
    hipLaunchKernelGGL(k_st_spmv_r<false>, dim3(nbnd + nfast), dim3(PGX_ROWMAP_BLOCK), 0, st, L.nx, L.ny, L.n, g, nbnd, L.K, L.M, L.Dh, L.Dd4,
                       make_stconst(L), L.mask, alpha, xu, xp, remap, yu, yp);
}

__global__ void __launch_bounds__(256) k_pack_d4(int n, const dsten_t* __restrict__ Dh, double4* __restrict__ Dd4) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  Dd4[v] = make_double4(Dh[v], Dh[(size_t)n + v], Dh[2 * (size_t)n + v], Dh[3 * (size_t)n + v]);
}
void pgxk_pack_d4(hipStream_t st, const GridLevel& L) {
  hipLaunchKernelGGL(k_pack_d4, dim3((L.n + 255) / 256), dim3(256), 0, st, L.n, L.Dh, L.Dd4);
}

void pgxk_st_apply(hipStream_t st, int mode, const GridLevel& L, double alpha, const double* xu, const double* xp,
                   const double* bu, const double* bp, double omega, int first, double* yu, double* yp) {
  dim3 grid((L.n + PGX_BLOCK - 1) / PGX_BLOCK), block(PGX_BLOCK);
  StConst sc;
  for (int s = 0; s < 7; ++s) {
    sc.K[s] = L.Kc[s];
    sc.M[s] = L.Mc[s];
  }
  sc.uniform = L.uniform;
  if (mode == 0)
    hipLaunchKernelGGL(k_st_apply<0>, grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask, alpha, xu, xp,
                       bu, bp, omega, first, yu, yp);
  else if (mode == 1)
    hipLaunchKernelGGL(k_st_apply<1>, grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask, alpha, xu, xp,
                       bu, bp, omega, first, yu, yp);
  else
    hipLaunchKernelGGL(k_st_apply<2>, grid, block, 0, st, L.nx, L.ny, L.n, L.K, L.M, L.Dh, sc, L.mask, alpha, xu, xp,
                       bu, bp, omega, first, yu, yp);
}

// b_c = P^T r_f  (u rows of coarse Dirichlet vertices get 0: they are not unknowns of the coarse problem)
__global__ void __launch_bounds__(PGX_BLOCK) k_restrict(int nxf, int nyf, const double* __restrict__ ru,
                                                        const double* __restrict__ rp, int nxc, int nyc, int ncv,
                                                        const uint8_t* __restrict__ mask_c, double* __restrict__ bu,
                                                        double* __restrict__ bp) {
  const int C = blockIdx.x * blockDim.x + threadIdx.x;
  if (C >= ncv) return;
  constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
  constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
  const int sxc = nxc + 1, sxf = nxf + 1;
  const int I = C % sxc, J = C / sxc;
  double su = 0.0, sp = 0.0;
#pragma unroll
  for (int o = 0; o < 7; ++o) {
    const int ax = 2 * I + OX[o], ay = 2 * J + OY[o];
    if (ax < 0 || ax > nxf || ay < 0 || ay > nyf) continue;
    const size_t a = (size_t)ay * sxf + ax;
    const double w = pw(OX[o], OY[o]);
    su += w * ru[a];
    sp += w * rp[a];
  }
  bu[C] = mask_c[C] ? 0.0 : su;
  bp[C] = sp;
}
void pgxk_restrict(hipStream_t st, const GridLevel& f, const double* ru, const double* rp, const GridLevel& c,
                   double* bu, double* bp) {
  hipLaunchKernelGGL(k_restrict, dim3((c.n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, f.nx, f.ny, ru, rp,
                     c.nx, c.ny, c.n, c.mask, bu, bp);
}

// x_f += P x_c
__global__ void __launch_bounds__(PGX_BLOCK) k_prolong_add(int nxc, const double* __restrict__ cu,
                                                           const double* __restrict__ cp, int nxf, int nf,
                                                           double* __restrict__ xu, double* __restrict__ xp) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nf) return;
  const int sxf = nxf + 1, sxc = nxc + 1;
  const int i = v % sxf, j = v / sxf;
  const int i0 = i >> 1, j0 = j >> 1;
  const int io = i & 1, jo = j & 1;
  // (i,j) lies between coarse vertices (i0,j0) and (i0+io, j0+jo): identical, x-edge, y-edge or diagonal midpoint
  const int c0 = j0 * sxc + i0, c1 = (j0 + jo) * sxc + (i0 + io);
  xu[v] += 0.5 * (cu[c0] + cu[c1]);
  xp[v] += 0.5 * (cp[c0] + cp[c1]);
}
void pgxk_prolong_add(hipStream_t st, const GridLevel& c, const double* cu, const double* cp, const GridLevel& f,
                      double* xu, double* xp) {
  hipLaunchKernelGGL(k_prolong_add, dim3((f.n + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, c.nx, cu, cp,
                     f.nx, f.n, xu, xp);
}

// ------------------------------------------------------------------------------------------------
// Fused multigrid tail: ONE launch, ONE workgroup runs the complete V-cycle over all levels small enough
// that a kernel per sweep is pure launch latency (rocprof: 68 launches of ~3.4 us + ~3.7 us gaps per cycle
// before fusion).  Phases are separated by __syncthreads(); all data is L2/L1 resident.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void restrict_vertex(int C, int nxf, int nyf, const double* ru, const double* rp, int nxc,
                                                const uint8_t* __restrict__ mask_c, double* bu, double* bp) {
  constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
  constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
  const int sxc = nxc + 1, sxf = nxf + 1;
  const int I = C % sxc, J = C / sxc;
  double su = 0.0, sp = 0.0;
#pragma unroll
  for (int o = 0; o < 7; ++o) {
    const int ax = 2 * I + OX[o], ay = 2 * J + OY[o];
    if (ax < 0 || ax > nxf || ay < 0 || ay > nyf) continue;
    const size_t a = (size_t)ay * sxf + ax;
    const double w = pw(OX[o], OY[o]);
    su += w * ru[a];
    sp += w * rp[a];
  }
  bu[C] = mask_c[C] ? 0.0 : su;
  bp[C] = sp;
}

__device__ __forceinline__ void prolong_vertex(int v, int nxc, const double* cu, const double* cp, int nxf, double* xu,
                                               double* xp) {
  const int sxf = nxf + 1, sxc = nxc + 1;
  const int i = v % sxf, j = v / sxf;
  const int i0 = i >> 1, j0 = j >> 1;
  const int c0 = j0 * sxc + i0, c1 = (j0 + (j & 1)) * sxc + (i0 + (i & 1));
  xu[v] += 0.5 * (cu[c0] + cu[c1]);
  xp[v] += 0.5 * (cp[c0] + cp[c1]);
}

__global__ void __launch_bounds__(1024) k_mg_tail(TailArgs A) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* cu[PGX_TAIL_MAX];
  const double* cp[PGX_TAIL_MAX];
  // ---- down leg ----
  for (int l = 0; l < A.nlev; ++l) {
    const TailLevel& L = A.L[l];
    const bool last = (l + 1 == A.nlev);
    const int total = last ? A.coarse_sweeps : 2 * A.nu;
    const int now = last ? A.coarse_sweeps : A.nu;
    bool toA = (total % 2) == 1;
    const double *xu = nullptr, *xp = nullptr;
    for (int s = 0; s < now; ++s) {
      double* tu = toA ? L.xu : L.xu2;
      double* tp = toA ? L.xp : L.xp2;
      for (int v = tid; v < L.n; v += nt)
        st_vertex<2>(v, L.nx, L.ny, L.n, L.K, L.M, L.Dh, L.sc, L.mask, A.alpha, xu, xp, L.bu, L.bp, A.omega, s == 0, tu,
                     tp);
      __syncthreads();
      xu = tu;
      xp = tp;
      toA = !toA;
    }
    cu[l] = xu;
    cp[l] = xp;
    if (!last) {
      for (int v = tid; v < L.n; v += nt)
        st_vertex<1>(v, L.nx, L.ny, L.n, L.K, L.M, L.Dh, L.sc, L.mask, A.alpha, xu, xp, L.bu, L.bp, 0.0, 0, L.ru, L.rp);
      __syncthreads();
      const TailLevel& C = A.L[l + 1];
      for (int c = tid; c < C.n; c += nt) restrict_vertex(c, L.nx, L.ny, L.ru, L.rp, C.nx, C.mask, C.bu, C.bp);
      __syncthreads();
    }
  }
  // ---- up leg ----
  for (int l = A.nlev - 2; l >= 0; --l) {
    const TailLevel& L = A.L[l];
    const TailLevel& C = A.L[l + 1];
    double* xu = (double*)cu[l];
    double* xp = (double*)cp[l];
    for (int v = tid; v < L.n; v += nt) prolong_vertex(v, C.nx, cu[l + 1], cp[l + 1], L.nx, xu, xp);
    __syncthreads();
    // after nu sweeps of 2*nu the current buffer is A if nu is even, B if odd; keep alternating so the
    // final sweep lands in A (= L.xu / L.xp), exactly like the host-driven cycle
    bool toA = (xu != L.xu);
    const double *su = xu, *sp = xp;
    for (int s = 0; s < A.nu; ++s) {
      double* tu = toA ? L.xu : L.xu2;
      double* tp = toA ? L.xp : L.xp2;
      for (int v = tid; v < L.n; v += nt)
        st_vertex<2>(v, L.nx, L.ny, L.n, L.K, L.M, L.Dh, L.sc, L.mask, A.alpha, su, sp, L.bu, L.bp, A.omega, 0, tu, tp);
      __syncthreads();
      su = tu;
      sp = tp;
      toA = !toA;
    }
    cu[l] = su;
    cp[l] = sp;
  }
}

// LDS-resident tail (the one used when the top tail level has <= 2048 vertices): k_mg_tail above spends
// ~3 us per phase on L2 round trips (rocprof: 100 us per call).  Here every vector of every tail level
// lives in LDS (8 arrays x sum(n_l) doubles <= 96 KB) and each thread keeps the stencil coefficients of its
// (at most two) vertices in registers for all sweeps of a level visit, so a phase is LDS traffic + barrier.
__device__ __forceinline__ void tail_load(const TailLevel& L, int v, StCoef& c, unsigned& nbm) {
  const int sx = L.nx + 1;
  const int i = v % sx, j = v / sx;
  st_load_coef(v, i, j, L.nx, L.ny, L.n, L.K, L.M, L.Dh, L.sc, L.mask, c);
  const int off[7] = {0, 1, -1, sx, -sx, sx + 1, -sx - 1};
  nbm = 0;
#pragma unroll
  for (int s = 0; s < 7; ++s)
    if (c.ok[s] && L.mask[v + off[s]]) nbm |= 1u << s;
}

__device__ __forceinline__ void tail_gather(const StCoef& c, unsigned nbm, int v, int sx, const double* xu,
                                            const double* xp, double xun[7], double xpn[7]) {
  const int off[7] = {0, 1, -1, sx, -sx, sx + 1, -sx - 1};
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int nb = c.ok[s] ? v + off[s] : v;
    xun[s] = ((nbm >> s) & 1u) ? 0.0 : xu[nb];
    xpn[s] = xp[nb];
  }
}

__global__ void __launch_bounds__(512) k_mg_tail_lds(TailArgs A) {
  extern __shared__ double lds[];
  constexpr int NT = 512, VPT = 3;  // 8 waves -> 256-VGPR budget: 3 vertices x 21 coefficients stay in registers
  const int tid = threadIdx.x;
  int base[PGX_TAIL_MAX];
  {
    int acc = 0;
    for (int l = 0; l < A.nlev; ++l) {
      base[l] = acc;
      acc += 8 * A.L[l].n;
    }
  }
  // array k of level l:  0,1: x (buffer 0)   2,3: x (buffer 1)   4,5: b   6,7: r
#define TL(l, k) (lds + base[l] + (k) * A.L[l].n)
  int cur[PGX_TAIL_MAX];
  for (int v = tid; v < A.L[0].n; v += NT) {
    TL(0, 4)[v] = A.L[0].bu[v];
    TL(0, 5)[v] = A.L[0].bp[v];
  }
  __syncthreads();
  StCoef c[VPT];
  unsigned nbm[VPT];
  auto sweeps = [&](int l, int count, bool from_zero, int& buf) {
    const TailLevel& L = A.L[l];
    const int sx = L.nx + 1;
    for (int s = 0; s < count; ++s) {
      const int src = buf, dst = buf ^ 1;
      const double *xu = TL(l, 2 * src), *xp = TL(l, 2 * src + 1);
      double *yu = TL(l, 2 * dst), *yp = TL(l, 2 * dst + 1);
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        const int v = tid + k * NT;
        if (v < L.n) {
          double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
          if (!(from_zero && s == 0)) {
            double xun[7], xpn[7];
            tail_gather(c[k], nbm[k], v, sx, xu, xp, xun, xpn);
            st_rows(c[k], A.alpha, xun, xpn, au, ap);
            xur = xu[v];
            xpr = xp[v];
          }
          double ou, op;
          st_jacobi(c[k], A.alpha, A.omega, au, ap, xur, xpr, TL(l, 4)[v], TL(l, 5)[v], ou, op);
          yu[v] = ou;
          yp[v] = op;
        }
      }
      __syncthreads();
      buf = dst;
    }
  };
  // ---- down leg ----
  for (int l = 0; l < A.nlev; ++l) {
    const TailLevel& L = A.L[l];
    const bool last = (l + 1 == A.nlev);
#pragma unroll
    for (int k = 0; k < VPT; ++k)
      if (tid + k * NT < L.n) tail_load(L, tid + k * NT, c[k], nbm[k]);
    int buf = 0;
    sweeps(l, last ? A.coarse_sweeps : A.nu, true, buf);
    cur[l] = buf;
    if (!last) {
      const int sx = L.nx + 1;
      const double *xu = TL(l, 2 * buf), *xp = TL(l, 2 * buf + 1);
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        const int v = tid + k * NT;
        if (v < L.n) {
          double xun[7], xpn[7], au, ap;
          tail_gather(c[k], nbm[k], v, sx, xu, xp, xun, xpn);
          st_rows(c[k], A.alpha, xun, xpn, au, ap);
          if (c[k].rowbc) au = xu[v];
          TL(l, 6)[v] = TL(l, 4)[v] - au;
          TL(l, 7)[v] = TL(l, 5)[v] - ap;
        }
      }
      __syncthreads();
      const TailLevel& C = A.L[l + 1];
      for (int cv = tid; cv < C.n; cv += NT)
        restrict_vertex(cv, L.nx, L.ny, TL(l, 6), TL(l, 7), C.nx, C.mask, TL(l + 1, 4), TL(l + 1, 5));
      __syncthreads();
    }
  }
  // ---- up leg ----
  for (int l = A.nlev - 2; l >= 0; --l) {
    const TailLevel& L = A.L[l];
    const TailLevel& C = A.L[l + 1];
    int buf = cur[l];
    for (int v = tid; v < L.n; v += NT)
      prolong_vertex(v, C.nx, TL(l + 1, 2 * cur[l + 1]), TL(l + 1, 2 * cur[l + 1] + 1), L.nx, TL(l, 2 * buf),
                     TL(l, 2 * buf + 1));
#pragma unroll
    for (int k = 0; k < VPT; ++k)
      if (tid + k * NT < L.n) tail_load(L, tid + k * NT, c[k], nbm[k]);
    __syncthreads();
    sweeps(l, A.nu, false, buf);
    cur[l] = buf;
  }
  for (int v = tid; v < A.L[0].n; v += NT) {
    A.L[0].xu[v] = TL(0, 2 * cur[0])[v];
    A.L[0].xp[v] = TL(0, 2 * cur[0] + 1)[v];
  }
#undef TL
}

// ------------------------------------------------------------------------------------------------
// k_mg_tail2 (round 3): the same V-cycle of the tail levels, restructured around what the trace of k_mg_tail_lds showed - 91 us
// per call for ~15 us of arithmetic: nine level visits that each start with a round trip to L2 for the level's stencil
// coefficients (7 D links, mask flags, boundary K / M rows: ~2 us per visit), three vertices per thread on the largest level and
// eight scalar LDS arrays.  Here
//  * EVERYTHING the cycle reads is staged into LDS by one batch of global loads at the start (right-hand side of the first
//    level; D half-stencils, Dirichlet flags and the K / M rows of the boundary vertices of every level): one memory latency
//    per launch instead of one per level visit;
//  * 1024 threads, at most two vertices per thread, so a sweep of the 33^2 level is one pass;
//  * (u, psi) interleaved as double2 (half the LDS instructions) and the residual written into the idle half of the solution
//    ping-pong pair, which frees the LDS for the staged coefficients.
// Same arithmetic per vertex as k_mg_tail_lds (st_rows / st_jacobi).  Requires uniform interior stencils on every tail level
// (the host checks; otherwise the kernels above run).
struct Tail2Off {
  int x0, x1, b, d, km, mk;  // byte offsets of a level's arrays in the dynamic LDS
  int vbase;                 // first index of the level in the concatenated vertex list of all tail levels
};
__host__ __device__ inline int tail2_nbnd(int nx, int ny) { return 2 * (nx + 1) + 2 * (ny - 1); }
__host__ __device__ inline size_t tail2_layout(const TailArgs& A, Tail2Off* off) {
  size_t acc = 0;
  int vb = 0;
  for (int l = 0; l < A.nlev; ++l) {
    const size_t n = (size_t)A.L[l].n;
    Tail2Off o;
    o.x0 = (int)acc;
    acc += 16 * n;
    o.x1 = (int)acc;
    acc += 16 * n;
    o.b = (int)acc;
    acc += 16 * n;
    o.d = (int)acc;
    acc += 32 * n;
    o.km = (int)acc;
    acc += (size_t)14 * 8 * tail2_nbnd(A.L[l].nx, A.L[l].ny);
    o.mk = (int)acc;
    acc += (n + 15) & ~(size_t)15;
    o.vbase = vb;
    vb += (int)n;
    if (off) off[l] = o;
  }
  return acc;
}

__device__ __forceinline__ int tail2_bidx(int i, int j, int nx, int ny) {
  return j == 0 ? i : (j == ny ? (nx + 1) + i : (i == 0 ? 2 * (nx + 1) + (j - 1) : 2 * (nx + 1) + (ny - 1) + (j - 1)));
}
__device__ __forceinline__ void tail2_bvertex(int t, int nx, int ny, int& i, int& j) {
  if (t < nx + 1) {
    i = t;
    j = 0;
  } else if (t < 2 * (nx + 1)) {
    i = t - (nx + 1);
    j = ny;
  } else if (t < 2 * (nx + 1) + (ny - 1)) {
    i = 0;
    j = t - 2 * (nx + 1) + 1;
  } else {
    i = nx;
    j = t - 2 * (nx + 1) - (ny - 1) + 1;
  }
}

__global__ void __launch_bounds__(512) k_mg_tail2(TailArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  constexpr int NT = 512, SV = 3, IV = 2;  // 8 waves: 256 VGPRs per thread keep every coefficient of a level visit in registers
  const int tid = threadIdx.x;
  // level tables live in LDS: a private array indexed by the (runtime) level would sit in scratch memory
  __shared__ Tail2Off off[PGX_TAIL_MAX];
  __shared__ int cur[PGX_TAIL_MAX];
  if (tid == 0) tail2_layout(A, off);
  __syncthreads();
  // ---- stage: ONE batch of global loads.  The vertices of all levels form one list (1493 entries for 33^2 ... 3^2); a thread
  // takes entries tid, tid + 512, tid + 1024 and has every load of both in flight before the first LDS store (a loop over the levels
  // would pay one memory latency per level: 10 of the first version's 63 us).
  {
    int sl[SV], sv[SV];
    double d[SV][4];
    uint8_t m[SV] = {};
    int total = 0;
    for (int l = 0; l < A.nlev; ++l) total += A.L[l].n;
#pragma unroll
    for (int k = 0; k < SV; ++k) {
      const int e = tid + k * NT;
      sl[k] = -1;
      sv[k] = 0;
      if (e < total) {
        int l = 0;
        while (l + 1 < A.nlev && e >= off[l + 1].vbase) ++l;
        sl[k] = l;
        sv[k] = e - off[l].vbase;
        const TailLevel& L = A.L[l];
        const size_t n = (size_t)L.n;
        d[k][0] = L.Dh[sv[k]];
        d[k][1] = L.Dh[n + sv[k]];
        d[k][2] = L.Dh[2 * n + sv[k]];
        d[k][3] = L.Dh[3 * n + sv[k]];
        m[k] = L.mask[sv[k]];
      }
    }
    // boundary rows: entry e of the concatenated boundary lists, one per thread (248 entries in all)
    int btotal = 0;
    for (int l = 0; l < A.nlev; ++l) btotal += tail2_nbnd(A.L[l].nx, A.L[l].ny);
    double kq[14];
    int bl = -1, bt = 0;
    if (tid < btotal) {
      int l = 0, base = 0;
      while (l + 1 < A.nlev && tid >= base + tail2_nbnd(A.L[l].nx, A.L[l].ny)) {
        base += tail2_nbnd(A.L[l].nx, A.L[l].ny);
        ++l;
      }
      bl = l;
      bt = tid - base;
      const TailLevel& L = A.L[l];
      int i, j;
      tail2_bvertex(bt, L.nx, L.ny, i, j);
      const size_t v = (size_t)j * (L.nx + 1) + i, n = (size_t)L.n;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        kq[s] = L.K[s * n + v];
        kq[7 + s] = L.M[s * n + v];
      }
    }
    double2 b0[SV];
#pragma unroll
    for (int k = 0; k < SV; ++k) {
      const int v = tid + k * NT;
      if (v < A.L[0].n) b0[k] = make_double2(A.L[0].bu[v], A.L[0].bp[v]);
    }
    // ---- all loads are in flight; now the stores ----
#pragma unroll
    for (int k = 0; k < SV; ++k)
      if (sl[k] >= 0) {
        const int l = sl[k], n = A.L[l].n, v = sv[k];
        double* D = (double*)(lds2 + off[l].d);
        D[v] = d[k][0];
        D[n + v] = d[k][1];
        D[2 * n + v] = d[k][2];
        D[3 * n + v] = d[k][3];
        (lds2 + off[l].mk)[v] = m[k];
      }
    if (bl >= 0) {
      double* q = (double*)(lds2 + off[bl].km) + 14 * bt;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        q[s] = A.alpha * kq[s];  // premultiplied
        q[7 + s] = kq[7 + s];
      }
    }
#pragma unroll
    for (int k = 0; k < SV; ++k) {
      const int v = tid + k * NT;
      if (v < A.L[0].n) ((double2*)(lds2 + off[0].b))[v] = b0[k];
    }
    // (host: every level's boundary list and the list of all levels' boundary vertices fit 1024 threads)
  }
  __syncthreads();

  // Vertex -> thread mapping of a level visit: thread t owns INTERIOR vertices t and t + 512 (uniform K / M stencils, pair-summed
  // and premultiplied on the host: scalar registers; the 7 D links and the 2x2 Jacobi block of each sit in vector registers), the
  // LAST 2(nx+1)+2(ny-1) threads own the BOUNDARY vertices (their whole rows in registers as well) - other waves than the
  // interior ones wherever the level is small, so the two instruction streams overlap.  The PMC profile of the first
  // version (profiles/r03_tail_pmc.json) showed the kernel bound by the dependent instruction stream of the one to four waves
  // that work on the small levels (76 % of all wave cycles waiting at barriers), hence: the damped inverse of the 2x2 vertex
  // block is formed ONCE per level visit (four multiply-adds per sweep instead of a reciprocal and ~25 instructions) and the row
  // sums run in independent chains.
  // Dirichlet columns need no masking here: on every level of the tail b_u is 0 on Dirichlet rows (the restriction masks it) and
  // the cycle starts from 0, so u stays exactly 0 there (the host requires `interior_free` levels, i.e. Dirichlet dofs on the
  // boundary only, whose interpolation parents are Dirichlet as well).
  double cd[IV][7];  // D links of the thread's (up to two) interior vertices
  double ig[IV][4];  // omega * inverse of the vertex block [[aK0, M0], [M0, -D0]] (identity row for a Dirichlet u)
  int irow[IV] = {0, 0}, iv[IV] = {-1, -1};
  double bg[4], ck[7], cm[7], bd[7];
  unsigned bok = 0;  // bit s: link s stays inside the grid
  int brow = 0, bv = -1, bt = 0;
  int xoff[2] = {0, 0}, boff = 0, doff = 0, kmoff = 0;  // LDS offsets of the current level (registers, not the LDS table)
  auto jac_block = [&](double a, double bm, double dd, int rowbc, double g[4]) {
    double om_u = A.omega;
    if (rowbc) {
      a = 1.0;
      bm = 0.0;
      om_u = 1.0;
    }
    const double det = -a * dd - bm * bm;
    g[0] = g[1] = g[2] = g[3] = 0.0;
    if (det != 0.0) {
      const double r = 1.0 / det;
      g[0] = om_u * (-dd * r);
      g[1] = om_u * (-bm * r);
      g[2] = A.omega * (-bm * r);
      g[3] = A.omega * (a * r);
    } else if (rowbc) {
      g[0] = 1.0;
    }
  };
  auto load_level = [&](int l) {
    const TailLevel& L = A.L[l];
    const int nx = L.nx, ny = L.ny, n = L.n, sx = nx + 1;
    xoff[0] = off[l].x0;
    xoff[1] = off[l].x1;
    boff = off[l].b;
    doff = off[l].d;
    kmoff = off[l].km;
    const double* D = (const double*)(lds2 + doff);
    const uint8_t* mk = lds2 + off[l].mk;
    bv = -1;
    const int nint = (nx > 1 && ny > 1) ? (nx - 1) * (ny - 1) : 0;
    const int nb = tail2_nbnd(nx, ny);
#pragma unroll
    for (int k = 0; k < IV; ++k) {
      iv[k] = -1;
      const int t = tid + k * NT;
      if (t < nint) {
        const int i = 1 + t % (nx - 1), j = 1 + t / (nx - 1);
        const int v = j * sx + i;
        iv[k] = v;
        cd[k][0] = D[v];
        cd[k][1] = D[n + v];
        cd[k][2] = D[n + v - 1];
        cd[k][3] = D[2 * n + v];
        cd[k][4] = D[2 * n + v - sx];
        cd[k][5] = D[3 * n + v];
        cd[k][6] = D[3 * n + v - sx - 1];
        irow[k] = mk[v];
        jac_block(L.ic[0], L.ic[4], cd[k][0], irow[k], ig[k]);
      }
    }
    if (tid >= NT - nb) {
      int i, j;
      const int t = tid - (NT - nb);
      tail2_bvertex(t, nx, ny, i, j);
      const int v = j * sx + i;
      bv = v;
      bt = t;
      bok = 1u | (i < nx ? 2u : 0u) | (i > 0 ? 4u : 0u) | (j < ny ? 8u : 0u) | (j > 0 ? 16u : 0u) | ((i < nx && j < ny) ? 32u : 0u) |
            ((i > 0 && j > 0) ? 64u : 0u);
      brow = mk[v];
      const double* q = (const double*)(lds2 + kmoff) + 14 * t;
      jac_block(q[0], q[7], D[v], brow, bg);
      const int dsl[7] = {v, n + v, n + v - 1, 2 * n + v, 2 * n + v - sx, 3 * n + v, 3 * n + v - sx - 1};
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        ck[s] = q[s];
        cm[s] = q[7 + s];
        bd[s] = ((bok >> s) & 1u) ? D[dsl[s]] : 0.0;
      }
    }
  };
  // one vertex: mode 0 = Jacobi update (from_zero: iterate taken as 0), mode 1 = residual
  auto interior_vertex = [&](int k, const TailLevel& L, int mode, bool from_zero, const double2* X, const double2* B, double2* Y) {
    const int v = iv[k], sx = L.nx + 1;
    double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
    if (!from_zero) {
      const double2 x0 = X[v], x1 = X[v + 1], x2 = X[v - 1], x3 = X[v + sx], x4 = X[v - sx], x5 = X[v + sx + 1], x6 = X[v - sx - 1];
      const double u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
      const double p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
      au = (L.ic[0] * x0.x + L.ic[1] * u12) + (L.ic[2] * u34 + L.ic[3] * u56) +
           ((L.ic[4] * x0.y + L.ic[5] * p12) + (L.ic[6] * p34 + L.ic[7] * p56));
      ap = ((L.ic[4] * x0.x + L.ic[5] * u12) + (L.ic[6] * u34 + L.ic[7] * u56)) -
           (((cd[k][0] * x0.y + cd[k][1] * x1.y) + (cd[k][2] * x2.y + cd[k][3] * x3.y)) +
            ((cd[k][4] * x4.y + cd[k][5] * x5.y) + cd[k][6] * x6.y));
      xur = x0.x;
      xpr = x0.y;
    }
    const double2 bq = B[v];
    if (irow[k]) au = xur;
    const double su = bq.x - au, sp = bq.y - ap;
    if (mode == 1) {
      Y[v] = make_double2(su, sp);
      return;
    }
    Y[v] = make_double2(xur + fma(ig[k][0], su, ig[k][1] * sp), xpr + fma(ig[k][2], su, ig[k][3] * sp));
  };
  auto boundary_vertex = [&](const TailLevel& L, int mode, bool from_zero, const double2* X, const double2* B, double2* Y) {
    const int v = bv, sx = L.nx + 1;
    const int o7[7] = {0, 1, -1, sx, -sx, sx + 1, -sx - 1};
    double au = 0.0, ap = 0.0, xur = 0.0, xpr = 0.0;
    if (!from_zero) {
      double2 t[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) t[s] = X[((bok >> s) & 1u) ? v + o7[s] : v];
      double au1 = 0.0, ap1 = 0.0;
#pragma unroll
      for (int s = 0; s < 7; ++s) {  // links that leave the grid carry zero coefficients
        if (s & 1) {
          au1 = fma(ck[s], t[s].x, fma(cm[s], t[s].y, au1));
          ap1 = fma(cm[s], t[s].x, fma(-bd[s], t[s].y, ap1));
        } else {
          au = fma(ck[s], t[s].x, fma(cm[s], t[s].y, au));
          ap = fma(cm[s], t[s].x, fma(-bd[s], t[s].y, ap));
        }
      }
      au += au1;
      ap += ap1;
      xur = t[0].x;
      xpr = t[0].y;
    }
    const double2 bq = B[v];
    if (brow) au = xur;
    const double su = bq.x - au, sp = bq.y - ap;
    if (mode == 1) {
      Y[v] = make_double2(su, sp);
      return;
    }
    Y[v] = make_double2(xur + fma(bg[0], su, bg[1] * sp), xpr + fma(bg[2], su, bg[3] * sp));
  };
  auto sweeps = [&](int l, int count, bool from_zero, int& buf) {
    const TailLevel& L = A.L[l];
    const double2* B = (const double2*)(lds2 + boff);
    for (int s = 0; s < count; ++s) {
      const double2* X = (const double2*)(lds2 + xoff[buf]);
      double2* Y = (double2*)(lds2 + xoff[buf ^ 1]);
      const bool fz = from_zero && s == 0;
#pragma unroll
      for (int k = 0; k < IV; ++k)
        if (iv[k] >= 0) interior_vertex(k, L, 0, fz, X, B, Y);
      if (bv >= 0) boundary_vertex(L, 0, fz, X, B, Y);
      __syncthreads();
      buf ^= 1;
    }
  };
#define T2X(l, k) ((double2*)(lds2 + ((k) ? off[l].x1 : off[l].x0)))
#define T2B(l) ((double2*)(lds2 + off[l].b))
#define T2M(l) ((uint8_t*)(lds2 + off[l].mk))
  // ---- down leg ----
  for (int l = 0; l < A.nlev; ++l) {
    const TailLevel& L = A.L[l];
    const bool last = (l + 1 == A.nlev);
    load_level(l);
    int buf = 0;
    sweeps(l, last ? A.coarse_sweeps : A.nu, true, buf);
    if (tid == 0) cur[l] = buf;  // read again on the way up, many barriers later
    if (last) __syncthreads();   // the coarsest level has no barrier after this write: the up leg / the copy-out read cur[l] next
    if (!last) {
      const double2* X = T2X(l, buf);
      double2* R = T2X(l, buf ^ 1);  // the idle half of the ping-pong pair holds the residual until it is restricted
      const double2* B = T2B(l);
#pragma unroll
      for (int k = 0; k < IV; ++k)
        if (iv[k] >= 0) interior_vertex(k, L, 1, false, X, B, R);
      if (bv >= 0) boundary_vertex(L, 1, false, X, B, R);
      __syncthreads();
      const TailLevel& C = A.L[l + 1];
      const int sxc = C.nx + 1, sxf = L.nx + 1;
      const uint8_t* mkc = T2M(l + 1);
      double2* Bc = T2B(l + 1);
      constexpr int OX[7] = {0, 1, -1, 0, 0, 1, -1};
      constexpr int OY[7] = {0, 0, 0, 1, -1, 1, -1};
      for (int cv = tid; cv < C.n; cv += NT) {
        const int I = cv % sxc, J = cv / sxc;
        double su = 0.0, sp = 0.0;
#pragma unroll
        for (int o = 0; o < 7; ++o) {
          const int ax = 2 * I + OX[o], ay = 2 * J + OY[o];
          if (ax < 0 || ax > L.nx || ay < 0 || ay > L.ny) continue;
          const double2 r = R[ay * sxf + ax];
          const double w = pw(OX[o], OY[o]);
          su += w * r.x;
          sp += w * r.y;
        }
        Bc[cv] = make_double2(mkc[cv] ? 0.0 : su, sp);
      }
      __syncthreads();
    }
  }
  // ---- up leg ----
  for (int l = A.nlev - 2; l >= 0; --l) {
    const TailLevel& L = A.L[l];
    const TailLevel& C = A.L[l + 1];
    int buf = cur[l];
    {
      double2* X = T2X(l, buf);
      const double2* Xc = T2X(l + 1, cur[l + 1]);
      const int sxf = L.nx + 1, sxc = C.nx + 1;
      for (int v = tid; v < L.n; v += NT) {
        const int i = v % sxf, j = v / sxf;
        const int i0 = i >> 1, j0 = j >> 1;
        const double2 a = Xc[j0 * sxc + i0], bq = Xc[(j0 + (j & 1)) * sxc + (i0 + (i & 1))];
        double2 x = X[v];
        x.x += 0.5 * (a.x + bq.x);
        x.y += 0.5 * (a.y + bq.y);
        X[v] = x;
      }
    }
    load_level(l);
    __syncthreads();
    sweeps(l, A.nu, false, buf);
    if (tid == 0) cur[l] = buf;
    __syncthreads();
  }
  {
    const double2* X = T2X(0, cur[0]);
    for (int v = tid; v < A.L[0].n; v += NT) {
      const double2 x = X[v];
      A.L[0].xu[v] = x.x;
      A.L[0].xp[v] = x.y;
    }
  }
#undef T2X
#undef T2B
#undef T2M
}

static int g_tail2 = 1;  // pgx_tuning: 0 = the round-2 kernels (A/B)
void pgxk_mg_tail_select(int v) { g_tail2 = v; }

void pgxk_mg_tail(hipStream_t st, const TailArgs& A0) {
  TailArgs A = A0;
  if (g_tail2) {
    bool ok = true;  // uniform interior stencils, Dirichlet dofs on the boundary only, every vertex list fits the 512 threads' slots
    int ntot = 0, nbtot = 0;
    for (int l = 0; l < A.nlev; ++l) {
      TailLevel& T = A.L[l];
      ok = ok && T.sc.uniform && T.interior_free && T.nx >= 1 && T.ny >= 1 && (T.nx - 1) * (T.ny - 1) <= 1024 &&
           tail2_nbnd(T.nx, T.ny) <= 512;
      ntot += T.n;
      nbtot += tail2_nbnd(T.nx, T.ny);
      // interior stencils of this launch: alpha K and M, symmetric link pairs pre-summed (scalar registers in the kernel)
      T.ic[0] = A.alpha * T.sc.K[0];
      T.ic[1] = A.alpha * 0.5 * (T.sc.K[1] + T.sc.K[2]);
      T.ic[2] = A.alpha * 0.5 * (T.sc.K[3] + T.sc.K[4]);
      T.ic[3] = A.alpha * 0.5 * (T.sc.K[5] + T.sc.K[6]);
      T.ic[4] = T.sc.M[0];
      T.ic[5] = 0.5 * (T.sc.M[1] + T.sc.M[2]);
      T.ic[6] = 0.5 * (T.sc.M[3] + T.sc.M[4]);
      T.ic[7] = 0.5 * (T.sc.M[5] + T.sc.M[6]);
    }
    ok = ok && ntot <= 3 * 512 && nbtot <= 512;
    const size_t need = tail2_layout(A, nullptr);
    if (ok && need <= 159 * 1024) {  // + 300 B of static LDS (level tables) under the 160 KB of a CU
      if (first_use_on_device(3))
        hipFuncSetAttribute((const void*)k_mg_tail2, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
      hipLaunchKernelGGL(k_mg_tail2, dim3(1), dim3(512), need, st, A);
      return;
    }
  }
  size_t total = 0;
  for (int l = 0; l < A.nlev; ++l) total += (size_t)8 * A.L[l].n * sizeof(double);
  if (A.L[0].n <= 1536 && total <= 150 * 1024) {
    if (first_use_on_device(2))
      hipFuncSetAttribute((const void*)k_mg_tail_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_mg_tail_lds, dim3(1), dim3(512), total, st, A);
  } else {
    hipLaunchKernelGGL(k_mg_tail, dim3(1), dim3(1024), 0, st, A);
  }
}
