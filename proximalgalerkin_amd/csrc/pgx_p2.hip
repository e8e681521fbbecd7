// P2 (quadratic Lagrange) element kernels for gfx950: obstacle_pg.py `-p 2` (obstacle_pg.py:68-70,288).
// Geometry: affine (from the cell's three vertices) or, since round 5, ISOPARAMETRIC of order 2 - every kernel takes an optional
// `geo` table [cell][quadrature point][5] = |det J|, J^-1 row-major of the quadratic cell map at that point (host:
// fem.Mesh.geometry_at; the reference's own meshes are gmsh meshes of element order 2, generate_mesh_gmsh.py:30-33) and then
// evaluates weights and physical gradients per point.  Dofs per field are [vertices | edge midpoints], local edge i opposite local
// vertex i.  Same structure as the P1 kernels of pgx_kernels.hip:
//   k_residual_p2        cell-parallel element residual, fp64 HW atomics
//   k_fill_rows_p2<MODE> row-parallel owner-computes fill of the P2 K / M / D(psi) CSR blocks (LDS-staged)
//   k_fill_rows_p1_Dp2   D(psi_P2) in the P1 basis == Galerkin coarse operator T^T D_P2 T (P1 c P2, same
//                        quadrature), feeding the P1 multigrid hierarchy without any sparse triple product
//   k_p2_restrict / k_p2_prolong_add   transfers T^T, T between the P2 space and its P1 subspace
#include "pgx_internal.h"

#define WAVE 64

struct Geom2 {
  double adet;
  double iJ[2][2];  // inverse Jacobian: G_a[d] = sum_k dN_a[k] * iJ[k][d]
};

__device__ __forceinline__ Geom2 geom2(const double* __restrict__ coords, int v0, int v1, int v2) {
  const double x0 = coords[2 * v0], y0 = coords[2 * v0 + 1];
  const double J00 = coords[2 * v1] - x0, J01 = coords[2 * v2] - x0;
  const double J10 = coords[2 * v1 + 1] - y0, J11 = coords[2 * v2 + 1] - y0;
  const double det = J00 * J11 - J01 * J10, inv = 1.0 / det;
  Geom2 g;
  g.adet = fabs(det);
  g.iJ[0][0] = J11 * inv;
  g.iJ[0][1] = -J01 * inv;
  g.iJ[1][0] = -J10 * inv;
  g.iJ[1][1] = J00 * inv;
  return g;
}

__device__ __forceinline__ Geom2 geom2q(const double* __restrict__ geo, size_t cq) {
  const double* p = geo + 5 * cq;
  Geom2 g;
  g.adet = p[0];
  g.iJ[0][0] = p[1];
  g.iJ[0][1] = p[2];
  g.iJ[1][0] = p[3];
  g.iJ[1][1] = p[4];
  return g;
}

__global__ void __launch_bounds__(PGX_BLOCK) k_bphi_p2(int nc, const int32_t* __restrict__ cdofs,
                                                       const double* __restrict__ coords,
                                                       const double* __restrict__ phi_q, QuadTab2 q,
                                                       double* __restrict__ stash, const double* __restrict__ geo) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  int d[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) d[a] = cdofs[6 * c + a];
  const Geom2 g = geom2(coords, d[0], d[1], d[2]);
  double b[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < q.nq; ++k) {
    // (affine: the weight 1.0 here and |det J| once below - bit for bit the arithmetic of rounds 1-4)
    const double wp = (geo ? geo[5 * ((size_t)c * q.nq + k)] : 1.0) * (q.w[k] * phi_q[(size_t)c * q.nq + k]);
#pragma unroll
    for (int a = 0; a < 6; ++a) b[a] += wp * q.N[k][a];
  }
  const double sc = geo ? 1.0 : g.adet;
#pragma unroll
  for (int a = 0; a < 6; ++a) stash[8 * (size_t)c + a] = sc * b[a];  // index = the dof lists' (cell * 8 + a)
}
void pgxk_bphi_p2(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords, const double* phi_q,
                  QuadTab2 q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* bphi, const double* geo) {
  hipLaunchKernelGGL(k_bphi_p2, dim3((nc + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nc, cdofs, coords,
                     phi_q, q, stash, geo);
  pgxk_gather_ent(st, n, v2c_ptr, v2c_ent, stash, bphi);
}

// residual (obstacle_pg.py:116-124) for P2; BC contract identical to k_residual_p1
__global__ void __launch_bounds__(PGX_BLOCK) k_residual_p2(int nc, int n, const int32_t* __restrict__ cdofs,
                                                           const double* __restrict__ coords,
                                                           const uint8_t* __restrict__ mask,
                                                           const double* __restrict__ gbc,
                                                           const double* __restrict__ x,
                                                           const double* __restrict__ xk, double alpha, double f,
                                                           QuadTab2 q, double* __restrict__ stash,
                                                           const double* __restrict__ geo) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  int d[6];
  double u[6], p[6], dp[6], Fu[6], Fp[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    d[a] = cdofs[6 * c + a];
    u[a] = mask[d[a]] ? gbc[d[a]] : x[d[a]];
    p[a] = x[n + d[a]];
    dp[a] = p[a] - xk[n + d[a]];
    Fu[a] = 0.0;
    Fp[a] = 0.0;
  }
  const Geom2 g0 = geom2(coords, d[0], d[1], d[2]);
  for (int k = 0; k < q.nq; ++k) {
    const Geom2 g = geo ? geom2q(geo, (size_t)c * q.nq + k) : g0;
    double uq = 0, pq = 0, dq = 0, gx = 0, gy = 0;
    double G[6][2];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      G[a][0] = q.dN[k][a][0] * g.iJ[0][0] + q.dN[k][a][1] * g.iJ[1][0];
      G[a][1] = q.dN[k][a][0] * g.iJ[0][1] + q.dN[k][a][1] * g.iJ[1][1];
      uq += u[a] * q.N[k][a];
      pq += p[a] * q.N[k][a];
      dq += dp[a] * q.N[k][a];
      gx += u[a] * G[a][0];
      gy += u[a] * G[a][1];
    }
    const double wd = g.adet * q.w[k];
    const double e = exp(pq);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      Fu[a] += wd * (alpha * (gx * G[a][0] + gy * G[a][1]) + (dq - alpha * f) * q.N[k][a]);
      Fp[a] += wd * (uq - e) * q.N[k][a];
    }
  }
  // element vectors parked at the index the dof -> (cell, local dof) lists use; summed per dof by k_gather_ent (no atomics)
  double* su = stash + 8 * (size_t)c;
  double* sp = stash + 8 * ((size_t)nc + c);
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    su[a] = Fu[a];
    sp[a] = Fp[a];
  }
}
void pgxk_residual_p2_cells(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords,
                            const uint8_t* mask, const double* gbc, const double* x, const double* xk, double alpha,
                            double f, QuadTab2 q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* F,
                            const double* geo) {
  hipLaunchKernelGGL(k_residual_p2, dim3((nc + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nc, n, cdofs,
                     coords, mask, gbc, x, xk, alpha, f, q, stash, geo);
  pgxk_gather_ent(st, n, v2c_ptr, v2c_ent, stash, F);
  pgxk_gather_ent(st, n, v2c_ptr, v2c_ent, stash + 8 * (size_t)nc, F + n);
}

// row-parallel fill of a P2 scalar CSR block (same LDS-image scheme as k_fill_rows of pgx_kernels.hip)
// v2c_ent[k] = cell*8 + local dof a; v2c_pos[2k], v2c_pos[2k+1]: row positions of the cell's dofs 0..3 / 4..5
template <int MODE>
__global__ void __launch_bounds__(PGX_BLOCK) k_fill_rows_p2(int n, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ v2c_ptr,
                                                            const int32_t* __restrict__ v2c_ent,
                                                            const int32_t* __restrict__ v2c_pos,
                                                            const int32_t* __restrict__ cdofs,
                                                            const double* __restrict__ coords,
                                                            const double* __restrict__ psi, QuadTab2 q,
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
    for (int k = v2c_ptr[i]; k < v2c_ptr[i + 1]; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 3, a = e & 7;
      const unsigned p0 = (unsigned)v2c_pos[2 * k], p1 = (unsigned)v2c_pos[2 * k + 1];
      int d[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) d[b] = cdofs[6 * c + b];
      const Geom2 g0 = geom2(coords, d[0], d[1], d[2]);
      double ps[6];
      if (MODE == 2) {
#pragma unroll
        for (int b = 0; b < 6; ++b) ps[b] = psi[d[b]];
      }
      double r[6] = {0, 0, 0, 0, 0, 0};
      for (int k2 = 0; k2 < q.nq; ++k2) {
        const Geom2 g = geo ? geom2q(geo, (size_t)c * q.nq + k2) : g0;
        const double wd = g.adet * q.w[k2];
        if (MODE == 0) {
          const double Ga0 = q.dN[k2][a][0] * g.iJ[0][0] + q.dN[k2][a][1] * g.iJ[1][0];
          const double Ga1 = q.dN[k2][a][0] * g.iJ[0][1] + q.dN[k2][a][1] * g.iJ[1][1];
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            const double Gb0 = q.dN[k2][b][0] * g.iJ[0][0] + q.dN[k2][b][1] * g.iJ[1][0];
            const double Gb1 = q.dN[k2][b][0] * g.iJ[0][1] + q.dN[k2][b][1] * g.iJ[1][1];
            r[b] += wd * (Ga0 * Gb0 + Ga1 * Gb1);
          }
        } else {
          double wa = wd * q.N[k2][a];
          if (MODE == 2) {
            double pq = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) pq += ps[b] * q.N[k2][b];
            wa *= exp(pq);
          }
#pragma unroll
          for (int b = 0; b < 6; ++b) r[b] += wa * q.N[k2][b];
        }
      }
      row[p0 & 0xff] += r[0];
      row[(p0 >> 8) & 0xff] += r[1];
      row[(p0 >> 16) & 0xff] += r[2];
      row[(p0 >> 24) & 0xff] += r[3];
      row[p1 & 0xff] += r[4];
      row[(p1 >> 8) & 0xff] += r[5];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) out[base + k] = acc[k];
}
void pgxk_fill_rows_p2(hipStream_t st, int mode, int n, size_t lds_bytes, const int32_t* rowptr,
                       const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cdofs,
                       const double* coords, const double* psi, QuadTab2 q, double* out, const double* geo) {
  dim3 grid((n + PGX_BLOCK - 1) / PGX_BLOCK), block(PGX_BLOCK);
  if (mode == 0)
    hipLaunchKernelGGL(k_fill_rows_p2<0>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cdofs,
                       coords, psi, q, out, geo);
  else if (mode == 1)
    hipLaunchKernelGGL(k_fill_rows_p2<1>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cdofs,
                       coords, psi, q, out, geo);
  else
    hipLaunchKernelGGL(k_fill_rows_p2<2>, grid, block, lds_bytes, st, n, rowptr, v2c_ptr, v2c_ent, v2c_pos, cdofs,
                       coords, psi, q, out, geo);
}

// D in the P1 basis with psi a P2 function: rows of the P1 plan (vertex -> incident cells)
__global__ void __launch_bounds__(PGX_BLOCK) k_fill_rows_p1_Dp2(int nv, const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ v2c_ptr,
                                                                const int32_t* __restrict__ v2c_ent,
                                                                const int32_t* __restrict__ v2c_pos,
                                                                const int32_t* __restrict__ cdofs,
                                                                const double* __restrict__ coords,
                                                                const double* __restrict__ psi, QuadTab2 q,
                                                                double* __restrict__ out, const double* __restrict__ geo) {
  extern __shared__ double acc[];
  const int i0 = blockIdx.x * PGX_BLOCK;
  const int i = i0 + threadIdx.x;
  const int iend = min(i0 + PGX_BLOCK, nv);
  const int base = rowptr[i0];
  const int len = rowptr[iend] - base;
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) acc[k] = 0.0;
  __syncthreads();
  if (i < nv) {
    double* row = acc + (rowptr[i] - base);
    for (int k = v2c_ptr[i]; k < v2c_ptr[i + 1]; ++k) {
      const int e = v2c_ent[k];
      const int c = e >> 2, a = e & 3;
      const int pos = v2c_pos[k];
      int d[6];
      double ps[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        d[b] = cdofs[6 * c + b];
        ps[b] = psi[d[b]];
      }
      const Geom2 g = geom2(coords, d[0], d[1], d[2]);
      double r[3] = {0, 0, 0};
      for (int k2 = 0; k2 < q.nq; ++k2) {
        double pq = 0.0;
#pragma unroll
        for (int b = 0; b < 6; ++b) pq += ps[b] * q.N[k2][b];
        const double wa = (geo ? geo[5 * ((size_t)c * q.nq + k2)] : g.adet) * q.w[k2] * exp(pq) * q.L[k2][a];
        r[0] += wa * q.L[k2][0];
        r[1] += wa * q.L[k2][1];
        r[2] += wa * q.L[k2][2];
      }
      row[pos & 0xff] += r[0];
      row[(pos >> 8) & 0xff] += r[1];
      row[(pos >> 16) & 0xff] += r[2];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += PGX_BLOCK) out[base + k] = acc[k];
}
void pgxk_fill_rows_p1_Dp2(hipStream_t st, int nv, size_t lds_bytes, const int32_t* rowptr, const int32_t* v2c_ptr,
                           const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cdofs, const double* coords,
                           const double* psi, QuadTab2 q, double* out, const double* geo) {
  hipLaunchKernelGGL(k_fill_rows_p1_Dp2, dim3((nv + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), lds_bytes, st, nv,
                     rowptr, v2c_ptr, v2c_ent, v2c_pos, cdofs, coords, psi, q, out, geo);
}

// r1 = T^T r2 (P1 hat = P2 vertex function + 1/2 of the adjacent edge functions); u rows of Dirichlet
// vertices get 0.  v2e: CSR vertex -> incident edge dofs (already offset by nv).
__global__ void __launch_bounds__(PGX_BLOCK) k_p2_restrict(int nv, int n2, const int32_t* __restrict__ v2e_ptr,
                                                           const int32_t* __restrict__ v2e, const uint8_t* __restrict__ mask1,
                                                           const double* __restrict__ ru2, const double* __restrict__ rp2,
                                                           double* __restrict__ bu1, double* __restrict__ bp1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  double su = ru2[i], sp = rp2[i];
  for (int k = v2e_ptr[i]; k < v2e_ptr[i + 1]; ++k) {
    const int e = v2e[k];
    su += 0.5 * ru2[e];
    sp += 0.5 * rp2[e];
  }
  bu1[i] = mask1[i] ? 0.0 : su;
  bp1[i] = sp;
}
void pgxk_p2_restrict(hipStream_t st, int nv, int n2, const int32_t* v2e_ptr, const int32_t* v2e, const uint8_t* mask1,
                      const double* ru2, const double* rp2, double* bu1, double* bp1) {
  hipLaunchKernelGGL(k_p2_restrict, dim3((nv + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nv, n2, v2e_ptr,
                     v2e, mask1, ru2, rp2, bu1, bp1);
}

// x2 += T x1
__global__ void __launch_bounds__(PGX_BLOCK) k_p2_prolong_add(int nv, int n2, const int32_t* __restrict__ edge_ends,
                                                              const double* __restrict__ cu, const double* __restrict__ cp,
                                                              double* __restrict__ xu, double* __restrict__ xp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  if (i < nv) {
    xu[i] += cu[i];
    xp[i] += cp[i];
  } else {
    const int a = edge_ends[2 * (i - nv)], b = edge_ends[2 * (i - nv) + 1];
    xu[i] += 0.5 * (cu[a] + cu[b]);
    xp[i] += 0.5 * (cp[a] + cp[b]);
  }
}
void pgxk_p2_prolong_add(hipStream_t st, int nv, int n2, const int32_t* edge_ends, const double* cu, const double* cp,
                         double* xu, double* xp) {
  hipLaunchKernelGGL(k_p2_prolong_add, dim3((n2 + PGX_BLOCK - 1) / PGX_BLOCK), dim3(PGX_BLOCK), 0, st, nv, n2,
                     edge_ends, cu, cp, xu, xp);
}

// six observables for P2 (gradient varies inside the cell: everything by quadrature)
__device__ __forceinline__ double wave_sum2(double v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__global__ void __launch_bounds__(PGX_BLOCK) k_observables_p2(int nc, int n, const int32_t* __restrict__ cdofs,
                                                              const double* __restrict__ coords,
                                                              const double* __restrict__ x,
                                                              const double* __restrict__ xk, double alpha, double f,
                                                              QuadTab2 q, double* __restrict__ partials,
                                                              const double* __restrict__ geo) {
  __shared__ double sm[6][PGX_BLOCK / WAVE];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    int d[6];
    double u[6], p[6], uk[6], pk[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      d[a] = cdofs[6 * c + a];
      u[a] = x[d[a]];
      p[a] = x[n + d[a]];
      uk[a] = xk[d[a]];
      pk[a] = xk[n + d[a]];
    }
    const Geom2 g0 = geom2(coords, d[0], d[1], d[2]);
    for (int k = 0; k < q.nq; ++k) {
      const Geom2 g = geo ? geom2q(geo, (size_t)c * q.nq + k) : g0;
      double uq = 0, pq = 0, ukq = 0, pkq = 0, gx = 0, gy = 0, hx = 0, hy = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double G0 = q.dN[k][a][0] * g.iJ[0][0] + q.dN[k][a][1] * g.iJ[1][0];
        const double G1 = q.dN[k][a][0] * g.iJ[0][1] + q.dN[k][a][1] * g.iJ[1][1];
        uq += u[a] * q.N[k][a];
        pq += p[a] * q.N[k][a];
        ukq += uk[a] * q.N[k][a];
        pkq += pk[a] * q.N[k][a];
        gx += u[a] * G0;
        gy += u[a] * G1;
        hx += (u[a] - uk[a]) * G0;
        hy += (u[a] - uk[a]) * G1;
      }
      const double wd = g.adet * q.w[k];
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
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double r = wave_sum2(s[k]);
    if (lane == 0) sm[k][wid] = r;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double r = 0.0;
    for (int w = 0; w < PGX_BLOCK / WAVE; ++w) r += sm[threadIdx.x][w];
    partials[blockIdx.x * 6 + threadIdx.x] = r;
  }
}
void pgxk_observables_p2_cells(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords,
                               const double* x, const double* xk, double alpha, double f, QuadTab2 q, double* partials,
                               int nblocks, const double* geo) {
  hipLaunchKernelGGL(k_observables_p2, dim3(nblocks), dim3(PGX_BLOCK), 0, st, nc, n, cdofs, coords, x, xk, alpha, f, q,
                     partials, geo);
}
