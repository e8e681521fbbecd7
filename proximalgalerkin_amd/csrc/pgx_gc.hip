// pgx_gc.hip - example 06 (gradient constraint, vector latent variable) behind the C ABI of include/pgx_gc.h.
//
// Reference: examples/06_gradient_constraints/gradient_constraint_dolfinx.py (:38-46 spaces, :53 degree-10 measure,
// :100-107 residual, :108-131 NonlinearProblem + SNES options, :171-205 outer loop).  u in P2, psi in (P1)^2,
// x = [u | psi_x | psi_y].  Newton matrix [[alpha K, G^T],[G, -N(psi)]] lives in ONE mixed CSR array whose pattern and
// per-cell destination tables are built once on the host; K and G are assembled once on the device, every Newton step
// only rescales the K slots and re-assembles the 36 N entries per cell.  Linear solves: pgx_nd (sparse LU) + iterative
// refinement on the exact operator.
#include <cstring>

#include "../../include/pgx_gc.h"
#include "pgx_mixed.h"
#include "pgx_scatter.h"

#define GC_MAXQ 40
struct GcQuad {
  double X[GC_MAXQ], Y[GC_MAXQ], w[GC_MAXQ];
  int nq;
};

static thread_local std::string g_gc_error;

struct pgx_gc_handle : MixedBase {
  int nv = 0, nc = 0, n2 = 0;
  GcQuad Q{};
  double alpha = 1.0;
  double *coords = nullptr, *phi = nullptr, *f = nullptr, *gbc = nullptr;
  int32_t* cdofs = nullptr;
  uint8_t* mask = nullptr;
  uint8_t* kind = nullptr;
  double* Jc = nullptr;  // constant part of the Jacobian values (K and G slots), assembled once
  // deterministic assembly (pgx_scatter.h): element kernels park [slot * nc + cell] in `stash`, one thread per destination sums
  PgxScatter sc_res, sc_N;  // residual: 12 slots per cell -> dofs; N(psi): 36 slots per cell -> CSR positions
  double* stash = nullptr;  // [36 * nc]  (general degree: [4 NP^2 * nc])
  // general primal degree k = 3..8 (pgx_gc_create_general): NU / NP local nodes of the primal P_k / latent P_(k-1) space, the latent
  // cell dofs, and the basis tables at the quadrature points; n2 = primal dofs, nv = latent dofs per component
  int gen = 0, NU = 6, NP = 3;
  int32_t *cdofs_p = nullptr, *cells3 = nullptr;  // latent cell dofs; vertex triples (affine geometry)
  double *tNu = nullptr, *tdNu = nullptr, *tNp = nullptr;
  void residual_dev(const double* xin, double* Fout) override;
  void jacobian_dev(const double* xin) override;
};

extern "C" const char* pgx_gc_last_error(const pgx_gc_handle* h) { return h ? h->err.c_str() : g_gc_error.c_str(); }

#define GCHIP MXHIP
#define GCALLOC MXALLOC
using GcTimer = MxTimer;

// ------------------------------------------------------------------------------------------------------------------
// element kernels (one thread per cell)
// ------------------------------------------------------------------------------------------------------------------
struct GcGeom {
  double inv[2][2];  // J^{-1}
  double adet;
};

__device__ inline GcGeom gc_geom(const double* __restrict__ coords, const int32_t* __restrict__ cd) {
  const double x0 = coords[2 * cd[0]], y0 = coords[2 * cd[0] + 1];
  const double j00 = coords[2 * cd[1]] - x0, j10 = coords[2 * cd[1] + 1] - y0;
  const double j01 = coords[2 * cd[2]] - x0, j11 = coords[2 * cd[2] + 1] - y0;
  const double det = j00 * j11 - j01 * j10;
  GcGeom g;
  g.inv[0][0] = j11 / det;
  g.inv[0][1] = -j01 / det;
  g.inv[1][0] = -j10 / det;
  g.inv[1][1] = j00 / det;
  g.adet = fabs(det);
  return g;
}

// P1 values l[3], P2 values N[6] and PHYSICAL P2 gradients G[6][2] at reference point (X,Y); edge i opposite vertex i
__device__ inline void gc_tab(double X, double Y, const GcGeom& g, double l[3], double N[6], double G[6][2]) {
  l[0] = 1.0 - X - Y;
  l[1] = X;
  l[2] = Y;
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  double dN[6][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    N[i] = l[i] * (2.0 * l[i] - 1.0);
    dN[i][0] = (4.0 * l[i] - 1.0) * dl[i][0];
    dN[i][1] = (4.0 * l[i] - 1.0) * dl[i][1];
  }
  const int ej[3] = {1, 0, 0}, ek[3] = {2, 2, 1};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    N[3 + i] = 4.0 * l[ej[i]] * l[ek[i]];
    dN[3 + i][0] = 4.0 * (l[ej[i]] * dl[ek[i]][0] + l[ek[i]] * dl[ej[i]][0]);
    dN[3 + i][1] = 4.0 * (l[ej[i]] * dl[ek[i]][1] + l[ek[i]] * dl[ej[i]][1]);
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    G[a][0] = dN[a][0] * g.inv[0][0] + dN[a][1] * g.inv[1][0];
    G[a][1] = dN[a][0] * g.inv[0][1] + dN[a][1] * g.inv[1][1];
  }
}

__global__ __launch_bounds__(128) void k_gc_residual(int nc, int n2, int nv, const int32_t* __restrict__ cdofs,
                                                     const double* __restrict__ coords, const uint8_t* __restrict__ mask,
                                                     const double* __restrict__ gbc, const double* __restrict__ phi,
                                                     const double* __restrict__ f, const double* __restrict__ x,
                                                     const double* __restrict__ xk, double alpha, GcQuad Q,
                                                     double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cd = cdofs + 6 * (size_t)c;
  const GcGeom g = gc_geom(coords, cd);
  double u[6], ph[6], ff[6], px[3], py[3], dx0[3], dy0[3];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int d = cd[a];
    u[a] = mask[d] ? gbc[d] : x[d];
    ph[a] = phi[d];
    ff[a] = f[d];
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int v = cd[b];
    px[b] = x[n2 + v];
    py[b] = x[n2 + nv + v];
    dx0[b] = px[b] - xk[n2 + v];
    dy0[b] = py[b] - xk[n2 + nv + v];
  }
  double Ru[6] = {0, 0, 0, 0, 0, 0}, Rx[3] = {0, 0, 0}, Ry[3] = {0, 0, 0};
  for (int q = 0; q < Q.nq; ++q) {
    double l[3], N[6], G[6][2];
    gc_tab(Q.X[q], Q.Y[q], g, l, N, G);
    const double wd = Q.w[q] * g.adet;
    double gux = 0, guy = 0, phq = 0, fq = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      gux += u[a] * G[a][0];
      guy += u[a] * G[a][1];
      phq += ph[a] * N[a];
      fq += ff[a] * N[a];
    }
    const double pxq = px[0] * l[0] + px[1] * l[1] + px[2] * l[2];
    const double pyq = py[0] * l[0] + py[1] * l[1] + py[2] * l[2];
    const double dxq = dx0[0] * l[0] + dx0[1] * l[1] + dx0[2] * l[2];
    const double dyq = dy0[0] * l[0] + dy0[1] * l[1] + dy0[2] * l[2];
    const double s = sqrt(1.0 + pxq * pxq + pyq * pyq);
    const double vx = alpha * gux + dxq, vy = alpha * guy + dyq;
#pragma unroll
    for (int a = 0; a < 6; ++a) Ru[a] += wd * (vx * G[a][0] + vy * G[a][1] - alpha * fq * N[a]);
    const double rx = gux - phq * pxq / s, ry = guy - phq * pyq / s;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      Rx[b] += wd * l[b] * rx;
      Ry[b] += wd * l[b] * ry;
    }
  }
  // parked slot-major (coalesced across the cells of a wave); pgx_scatter sums them per dof in a fixed order
#pragma unroll
  for (int a = 0; a < 6; ++a) stash[(size_t)a * nc + c] = Ru[a];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    stash[(size_t)(6 + b) * nc + c] = Rx[b];
    stash[(size_t)(9 + b) * nc + c] = Ry[b];
  }
}

__global__ void k_gc_resid_bc(int n2, const uint8_t* __restrict__ mask, const double* __restrict__ gbc,
                              const double* __restrict__ x, double* __restrict__ F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2 && mask[i]) F[i] = x[i] - gbc[i];
}

// constant part, once: K (36 entries) and G, G^T (36 + 36) per cell, parked in a [108 * nc] stash
__global__ __launch_bounds__(128) void k_gc_const(int nc, const int32_t* __restrict__ cdofs, const double* __restrict__ coords,
                                                  GcQuad Q, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cd = cdofs + 6 * (size_t)c;
  const GcGeom g = gc_geom(coords, cd);
  double Ke[6][6], Ge[3][2][6];
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) Ke[a][b] = 0.0;
  for (int b = 0; b < 3; ++b)
    for (int d = 0; d < 2; ++d)
      for (int a = 0; a < 6; ++a) Ge[b][d][a] = 0.0;
  for (int q = 0; q < Q.nq; ++q) {
    double l[3], N[6], G[6][2];
    gc_tab(Q.X[q], Q.Y[q], g, l, N, G);
    const double wd = Q.w[q] * g.adet;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b < 6; ++b) Ke[a][b] += wd * (G[a][0] * G[b][0] + G[a][1] * G[b][1]);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        Ge[b][0][a] += wd * l[b] * G[a][0];
        Ge[b][1][a] += wd * l[b] * G[a][1];
      }
    }
  }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) stash[(size_t)(a * 6 + b) * nc + c] = Ke[a][b];
  for (int b = 0; b < 3; ++b)
    for (int d = 0; d < 2; ++d)
      for (int a = 0; a < 6; ++a) {
        const int e = 36 + ((b * 2 + d) * 6 + a) * 2;
        stash[(size_t)e * nc + c] = Ge[b][d][a];
        stash[(size_t)(e + 1) * nc + c] = Ge[b][d][a];
      }
}

// kind: 0 = K slot (scaled by alpha), 1 = G / G^T slot, 2 = N slot (accumulated by k_gc_jac_N), 3 = BC diagonal, 4 = zeroed by BCs
__global__ void k_gc_jac_init(int64_t nnz, const uint8_t* __restrict__ kind, const double* __restrict__ Jc, double alpha,
                              double* __restrict__ Jv) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int t = kind[k];
  Jv[k] = t == 0 ? alpha * Jc[k] : t == 1 ? Jc[k] : t == 3 ? 1.0 : 0.0;
}

__global__ __launch_bounds__(128) void k_gc_jac_N(int nc, int n2, int nv, const int32_t* __restrict__ cdofs,
                                                  const double* __restrict__ coords, const double* __restrict__ phi,
                                                  const double* __restrict__ x, GcQuad Q, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cd = cdofs + 6 * (size_t)c;
  const GcGeom g = gc_geom(coords, cd);
  double ph[6], px[3], py[3];
#pragma unroll
  for (int a = 0; a < 6; ++a) ph[a] = phi[cd[a]];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    px[b] = x[n2 + cd[b]];
    py[b] = x[n2 + nv + cd[b]];
  }
  double Nxx[3][3], Nxy[3][3], Nyy[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Nxx[a][b] = Nxy[a][b] = Nyy[a][b] = 0.0;
  for (int q = 0; q < Q.nq; ++q) {
    double l[3], N[6], G[6][2];
    gc_tab(Q.X[q], Q.Y[q], g, l, N, G);
    double phq = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) phq += ph[a] * N[a];
    const double pxq = px[0] * l[0] + px[1] * l[1] + px[2] * l[2];
    const double pyq = py[0] * l[0] + py[1] * l[1] + py[2] * l[2];
    const double s = sqrt(1.0 + pxq * pxq + pyq * pyq);
    const double s3 = s * s * s;
    const double wp = Q.w[q] * g.adet * phq;
    const double cxx = wp * (1.0 / s - pxq * pxq / s3), cxy = wp * (-pxq * pyq / s3), cyy = wp * (1.0 / s - pyq * pyq / s3);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const double ll = l[a] * l[b];
        Nxx[a][b] += cxx * ll;
        Nxy[a][b] += cxy * ll;
        Nyy[a][b] += cyy * ll;
      }
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const int e = (a * 3 + b) * 4;  // (c,d) = xx, xy, yx, yy
      stash[(size_t)e * nc + c] = Nxx[a][b];
      stash[(size_t)(e + 1) * nc + c] = Nxy[a][b];
      stash[(size_t)(e + 2) * nc + c] = Nxy[a][b];
      stash[(size_t)(e + 3) * nc + c] = Nyy[a][b];
    }
}

#define GC_RED MX_RED
// int (u - uk)^2 with the problem's quadrature: per-block partial sums over cells
__global__ __launch_bounds__(256) void k_gc_l2(int nc, const int32_t* __restrict__ cdofs, const double* __restrict__ coords,
                                               const double* __restrict__ x, const double* __restrict__ xk, GcQuad Q,
                                               double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < nc; c += GC_RED * 256) {
    const int32_t* cd = cdofs + 6 * (size_t)c;
    const GcGeom g = gc_geom(coords, cd);
    double d[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = x[cd[a]] - xk[cd[a]];
    for (int q = 0; q < Q.nq; ++q) {
      double l[3], N[6], G[6][2];
      gc_tab(Q.X[q], Q.Y[q], g, l, N, G);
      double v = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) v += d[a] * N[a];
      s += Q.w[q] * g.adet * v * v;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// ------------------------------------------------------------------------------------------------------------------
// general degree (primal P_k, latent (P_(k-1))^2, k <= 8): the same element integrals, TABLE-driven.  tNu [nq][NU] and tdNu
// [nq][NU][2] are the primal basis and its reference gradients at the quadrature points, tNp [nq][NP] the latent basis; cells are
// affine (geometry from the three vertices).  One thread per cell, runtime sizes, local arrays of the maximal size; slot-major
// stashes as above.  These kernels are not tuned: the factorisation dominates a Newton step by orders of magnitude.
// ------------------------------------------------------------------------------------------------------------------
// element sizes: triangles P8 / P7 have 45 / 36 local nodes, quadrilaterals Q8 / Q7 81 / 64 (pgx_gc.h: affine cells of either shape)
#define GCG_MAXU 81
#define GCG_MAXP 64
#define GCG_TRIU 45
#define GCG_TRIP 36
struct GcgArgs {
  int nc, n2, nv, NU, NP, nq;
  const int32_t *cells, *cdu, *cdp;
  const double *coords, *tNu, *tdNu, *tNp;
  double w[GC_MAXQ];
};

__device__ inline GcGeom gcg_geom(const GcgArgs& A, int c) { return gc_geom(A.coords, A.cells + 3 * (size_t)c); }

template <int MAXU, int MAXP>
__global__ __launch_bounds__(64) void k_gcg_residual(GcgArgs A, const uint8_t* __restrict__ mask, const double* __restrict__ gbc,
                                                     const double* __restrict__ phi, const double* __restrict__ f,
                                                     const double* __restrict__ x, const double* __restrict__ xk, double alpha,
                                                     double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.nc) return;
  const int NU = A.NU, NP = A.NP, nc = A.nc;
  const GcGeom g = gcg_geom(A, c);
  const int32_t* cu = A.cdu + (size_t)NU * c;
  const int32_t* cp = A.cdp + (size_t)NP * c;
  double u[MAXU], ph[MAXU], ff[MAXU], Ru[MAXU], Gx[MAXU], Gy[MAXU];
  double px[MAXP], py[MAXP], dx0[MAXP], dy0[MAXP], Rx[MAXP], Ry[MAXP];
  for (int a = 0; a < NU; ++a) {
    const int d = cu[a];
    u[a] = mask[d] ? gbc[d] : x[d];
    ph[a] = phi[d];
    ff[a] = f[d];
    Ru[a] = 0.0;
  }
  for (int b = 0; b < NP; ++b) {
    const int v = cp[b];
    px[b] = x[A.n2 + v];
    py[b] = x[A.n2 + A.nv + v];
    dx0[b] = px[b] - xk[A.n2 + v];
    dy0[b] = py[b] - xk[A.n2 + A.nv + v];
    Rx[b] = Ry[b] = 0.0;
  }
  for (int q = 0; q < A.nq; ++q) {
    const double* Nu = A.tNu + (size_t)q * NU;
    const double* dN = A.tdNu + (size_t)q * NU * 2;
    const double* Np = A.tNp + (size_t)q * NP;
    const double wd = A.w[q] * g.adet;
    double gux = 0, guy = 0, phq = 0, fq = 0;
    for (int a = 0; a < NU; ++a) {
      Gx[a] = dN[2 * a] * g.inv[0][0] + dN[2 * a + 1] * g.inv[1][0];
      Gy[a] = dN[2 * a] * g.inv[0][1] + dN[2 * a + 1] * g.inv[1][1];
      gux += u[a] * Gx[a];
      guy += u[a] * Gy[a];
      phq += ph[a] * Nu[a];
      fq += ff[a] * Nu[a];
    }
    double pxq = 0, pyq = 0, dxq = 0, dyq = 0;
    for (int b = 0; b < NP; ++b) {
      pxq += px[b] * Np[b];
      pyq += py[b] * Np[b];
      dxq += dx0[b] * Np[b];
      dyq += dy0[b] * Np[b];
    }
    const double s = sqrt(1.0 + pxq * pxq + pyq * pyq);
    const double vx = alpha * gux + dxq, vy = alpha * guy + dyq;
    for (int a = 0; a < NU; ++a) Ru[a] += wd * (vx * Gx[a] + vy * Gy[a] - alpha * fq * Nu[a]);
    const double rx = gux - phq * pxq / s, ry = guy - phq * pyq / s;
    for (int b = 0; b < NP; ++b) {
      Rx[b] += wd * Np[b] * rx;
      Ry[b] += wd * Np[b] * ry;
    }
  }
  for (int a = 0; a < NU; ++a) stash[(size_t)a * nc + c] = Ru[a];
  for (int b = 0; b < NP; ++b) {
    stash[(size_t)(NU + b) * nc + c] = Rx[b];
    stash[(size_t)(NU + NP + b) * nc + c] = Ry[b];
  }
}

// constant part: K (NU^2 slots), then G and G^T: slot NU^2 + ((b*2 + d)*NU + a)*2 (+1)
__global__ __launch_bounds__(64) void k_gcg_const(GcgArgs A, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.nc) return;
  const int NU = A.NU, NP = A.NP, nc = A.nc;
  const GcGeom g = gcg_geom(A, c);
  for (int a = 0; a < NU; ++a) {
    for (int b = 0; b < NU; ++b) {
      double acc = 0.0;
      for (int q = 0; q < A.nq; ++q) {
        const double* dN = A.tdNu + (size_t)q * NU * 2;
        const double ax = dN[2 * a] * g.inv[0][0] + dN[2 * a + 1] * g.inv[1][0], ay = dN[2 * a] * g.inv[0][1] + dN[2 * a + 1] * g.inv[1][1];
        const double bx = dN[2 * b] * g.inv[0][0] + dN[2 * b + 1] * g.inv[1][0], by = dN[2 * b] * g.inv[0][1] + dN[2 * b + 1] * g.inv[1][1];
        acc += A.w[q] * g.adet * (ax * bx + ay * by);
      }
      stash[(size_t)(a * NU + b) * nc + c] = acc;
    }
    for (int b = 0; b < NP; ++b) {
      double gx = 0.0, gy = 0.0;
      for (int q = 0; q < A.nq; ++q) {
        const double* dN = A.tdNu + (size_t)q * NU * 2;
        const double ax = dN[2 * a] * g.inv[0][0] + dN[2 * a + 1] * g.inv[1][0], ay = dN[2 * a] * g.inv[0][1] + dN[2 * a + 1] * g.inv[1][1];
        const double wl = A.w[q] * g.adet * A.tNp[(size_t)q * NP + b];
        gx += wl * ax;
        gy += wl * ay;
      }
      const size_t e0 = (size_t)NU * NU + ((size_t)(b * 2 + 0) * NU + a) * 2, e1 = (size_t)NU * NU + ((size_t)(b * 2 + 1) * NU + a) * 2;
      stash[e0 * nc + c] = gx;
      stash[(e0 + 1) * nc + c] = gx;
      stash[e1 * nc + c] = gy;
      stash[(e1 + 1) * nc + c] = gy;
    }
  }
}

// N(psi): slot (a*NP + b)*4 + {xx, xy, yx, yy}
__global__ __launch_bounds__(64) void k_gcg_jac_N(GcgArgs A, const double* __restrict__ phi, const double* __restrict__ x,
                                                  double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.nc) return;
  const int NU = A.NU, NP = A.NP, nc = A.nc;
  const GcGeom g = gcg_geom(A, c);
  const int32_t* cu = A.cdu + (size_t)NU * c;
  const int32_t* cp = A.cdp + (size_t)NP * c;
  double cxx[GC_MAXQ], cxy[GC_MAXQ], cyy[GC_MAXQ];
  for (int q = 0; q < A.nq; ++q) {
    const double* Nu = A.tNu + (size_t)q * NU;
    const double* Np = A.tNp + (size_t)q * NP;
    double phq = 0, pxq = 0, pyq = 0;
    for (int a = 0; a < NU; ++a) phq += phi[cu[a]] * Nu[a];
    for (int b = 0; b < NP; ++b) {
      pxq += x[A.n2 + cp[b]] * Np[b];
      pyq += x[A.n2 + A.nv + cp[b]] * Np[b];
    }
    const double s = sqrt(1.0 + pxq * pxq + pyq * pyq), s3 = s * s * s, wp = A.w[q] * g.adet * phq;
    cxx[q] = wp * (1.0 / s - pxq * pxq / s3);
    cxy[q] = wp * (-pxq * pyq / s3);
    cyy[q] = wp * (1.0 / s - pyq * pyq / s3);
  }
  for (int a = 0; a < NP; ++a)
    for (int b = 0; b < NP; ++b) {
      double nxx = 0, nxy = 0, nyy = 0;
      for (int q = 0; q < A.nq; ++q) {
        const double ll = A.tNp[(size_t)q * NP + a] * A.tNp[(size_t)q * NP + b];
        nxx += cxx[q] * ll;
        nxy += cxy[q] * ll;
        nyy += cyy[q] * ll;
      }
      const size_t e = (size_t)(a * NP + b) * 4;
      stash[e * nc + c] = nxx;
      stash[(e + 1) * nc + c] = nxy;
      stash[(e + 2) * nc + c] = nxy;
      stash[(e + 3) * nc + c] = nyy;
    }
}

__global__ __launch_bounds__(256) void k_gcg_l2(GcgArgs A, const double* __restrict__ x, const double* __restrict__ xk,
                                                double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < A.nc; c += MX_RED * 256) {
    const GcGeom g = gcg_geom(A, c);
    const int32_t* cu = A.cdu + (size_t)A.NU * c;
    for (int q = 0; q < A.nq; ++q) {
      const double* Nu = A.tNu + (size_t)q * A.NU;
      double v = 0;
      for (int a = 0; a < A.NU; ++a) v += (x[cu[a]] - xk[cu[a]]) * Nu[a];
      s += A.w[q] * g.adet * v * v;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

static GcgArgs gcg_args(const pgx_gc_handle* h);

// ------------------------------------------------------------------------------------------------------------------
// host: pattern, destination tables, create
// ------------------------------------------------------------------------------------------------------------------
extern "C" void pgx_gc_destroy(pgx_gc_handle* h) {
  if (!h) return;
  mx_release(h);
  delete h;
}

static int gc_create_impl(pgx_gc_handle* h, const pgx_mesh* m, const pgx_gc_problem* p, pgx_comm* comm) {
  h->comm = comm;
  const int nv = m->n_vertices, nc = m->n_cells, n2 = m->n_dofs;
  const int64_t ntot = (int64_t)n2 + 2 * (int64_t)nv;
  h->nv = nv, h->nc = nc, h->n2 = n2, h->ntot = ntot;
  h->Q.nq = p->nq;
  for (int q = 0; q < p->nq; ++q) h->Q.X[q] = p->qpts[2 * q], h->Q.Y[q] = p->qpts[2 * q + 1], h->Q.w[q] = p->qwts[q];
  const int32_t* cd = m->cell_dofs;
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 6; ++a) {
      const int v = cd[6 * (size_t)c + a];
      if (v < 0 || v >= n2 || (a < 3 && (v >= nv || v != m->cells[3 * (size_t)c + a])) || (a >= 3 && v < nv)) {
        h->err = "cell_dofs must be [vertex ids (== cells) | edge dofs >= n_vertices]";
        return PGX_EINVAL;
      }
    }
  std::vector<uint8_t> hmask(n2, 0);
  std::vector<double> hg(n2, 0.0);
  for (int k = 0; k < p->n_bc; ++k) {
    const int d = p->bc_dofs[k];
    if (d < 0 || d >= n2) {
      h->err = "bc dof out of range";
      return PGX_EINVAL;
    }
    hmask[d] = 1;
    hg[d] = p->bc_vals ? p->bc_vals[k] : 0.0;
  }
  // mixed dofs of a cell: 6 u, 3 psi_x, 3 psi_y
  auto mixed = [&](int c, int32_t md[12]) {
    for (int a = 0; a < 6; ++a) md[a] = cd[6 * (size_t)c + a];
    for (int b = 0; b < 3; ++b) md[6 + b] = n2 + cd[6 * (size_t)c + b], md[9 + b] = n2 + nv + cd[6 * (size_t)c + b];
  };
  // dof -> cells
  std::vector<int64_t> dptr(ntot + 1, 0);
  for (int c = 0; c < nc; ++c) {
    int32_t md[12];
    mixed(c, md);
    for (int a = 0; a < 12; ++a) dptr[md[a] + 1]++;
  }
  for (int64_t i = 0; i < ntot; ++i) dptr[i + 1] += dptr[i];
  std::vector<int32_t> dcell(dptr[ntot]);
  {
    std::vector<int64_t> fill(dptr.begin(), dptr.end() - 1);
    for (int c = 0; c < nc; ++c) {
      int32_t md[12];
      mixed(c, md);
      for (int a = 0; a < 12; ++a) dcell[fill[md[a]]++] = c;
    }
  }
  // pattern: row r couples to every mixed dof of its cells
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(ntot + 1, 0);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp;
    for (int64_t r = a; r < b; ++r) {
      tmp.clear();
      for (int64_t q = dptr[r]; q < dptr[r + 1]; ++q) {
        int32_t md[12];
        mixed(dcell[q], md);
        tmp.insert(tmp.end(), md, md + 12);
      }
      std::sort(tmp.begin(), tmp.end());
      rowptr[r + 1] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
    }
  });
  int64_t tot = 0;
  for (int64_t r = 0; r < ntot; ++r) {
    tot += rowptr[r + 1];
    if (tot > 0x7fffffff) {
      h->err = "mixed matrix exceeds int32 nnz";
      return PGX_EINVAL;
    }
    rowptr[r + 1] = (int32_t)tot;
  }
  h->nnz = tot;
  col.resize(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp;
    for (int64_t r = a; r < b; ++r) {
      tmp.clear();
      for (int64_t q = dptr[r]; q < dptr[r + 1]; ++q) {
        int32_t md[12];
        mixed(dcell[q], md);
        tmp.insert(tmp.end(), md, md + 12);
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      std::copy(tmp.begin(), tmp.end(), col.begin() + rowptr[r]);
    }
  });
  auto find = [&](int32_t r, int32_t c) -> int32_t {
    const int32_t* b = col.data() + rowptr[r];
    const int32_t* e = col.data() + rowptr[r + 1];
    return (int32_t)(std::lower_bound(b, e, c) - col.data());
  };
  // slot kinds
  std::vector<uint8_t> kind(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r)
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int32_t c = col[k];
        uint8_t t;
        if (r < n2 && c < n2)
          t = (hmask[r] || hmask[c]) ? ((r == c && hmask[r]) ? 3 : 4) : 0;
        else if (r < n2)
          t = hmask[r] ? 4 : 1;
        else if (c < n2)
          t = hmask[c] ? 4 : 1;
        else
          t = 2;
        kind[k] = t;
      }
  });
  // destination tables
  std::vector<int32_t> d108((size_t)nc * 108), d36((size_t)nc * 36), d12((size_t)nc * 12);
  mx_par_for(nc, [&](int64_t a0, int64_t b0) {
    for (int64_t c = a0; c < b0; ++c) {
      int32_t md[12];
      mixed((int)c, md);
      // slot-major like the stashes the element kernels write: table[slot * nc + cell]
      auto D = [&](int e) -> int32_t& { return d108[(size_t)e * nc + (size_t)c]; };
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) D(a * 6 + b) = find(md[a], md[b]);
      for (int b = 0; b < 3; ++b)
        for (int d = 0; d < 2; ++d)
          for (int a = 0; a < 6; ++a) {
            const int e = 36 + ((b * 2 + d) * 6 + a) * 2;
            const int32_t pr = md[6 + 3 * d + b];
            D(e) = find(pr, md[a]);
            D(e + 1) = find(md[a], pr);
          }
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          for (int cc = 0; cc < 2; ++cc)
            for (int d = 0; d < 2; ++d)
              d36[(size_t)((a * 3 + b) * 4 + cc * 2 + d) * nc + (size_t)c] = find(md[6 + 3 * cc + a], md[6 + 3 * d + b]);
      for (int a = 0; a < 12; ++a) d12[(size_t)a * nc + (size_t)c] = md[a];
    }
  });
  // node coordinates for the nested-dissection ordering: vertices, then edge midpoints
  std::vector<double> xy(2 * (size_t)n2, 0.0);
  std::copy(m->coords, m->coords + 2 * (size_t)nv, xy.begin());
  for (int c = 0; c < nc; ++c) {
    const int ej[3] = {1, 0, 0}, ek[3] = {2, 2, 1};
    for (int i = 0; i < 3; ++i) {
      const int e = cd[6 * (size_t)c + 3 + i], vj = cd[6 * (size_t)c + ej[i]], vk = cd[6 * (size_t)c + ek[i]];
      xy[2 * (size_t)e] = 0.5 * (m->coords[2 * (size_t)vj] + m->coords[2 * (size_t)vk]);
      xy[2 * (size_t)e + 1] = 0.5 * (m->coords[2 * (size_t)vj + 1] + m->coords[2 * (size_t)vk + 1]);
    }
  }
  std::vector<int32_t> nod(ntot);
  for (int i = 0; i < n2; ++i) nod[i] = i;
  for (int v = 0; v < nv; ++v) nod[n2 + v] = nod[(size_t)n2 + nv + v] = v;
  // device
  GCHIP(hipStreamCreate(&h->st));
  pgx_nd_matrix A{};
  A.n = ntot;
  A.rowptr = rowptr.data();
  A.col = col.data();
  A.n_nodes = n2;
  A.node_of_dof = nod.data();
  A.dim = 2;
  A.node_coords = xy.data();
  A.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) A.leaf_nodes = atoi(e);
  int rc = comm ? pgx_nd_create_dist(&A, comm, h->device, (void*)h->st, &h->lu) : pgx_nd_create(&A, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  // the Newton matrix [[alpha K, G^T], [G, -N(psi)]] is symmetric (indefinite): L D L^T in LU clothing, half the flops (pgx_nd.h)
  pgx_nd_set_symmetric(h->lu, 1);
  GCALLOC(h->coords, 2 * (size_t)nv);
  GCALLOC(h->cdofs, 6 * (size_t)nc);
  GCALLOC(h->mask, n2);
  GCALLOC(h->gbc, n2);
  GCALLOC(h->phi, n2);
  GCALLOC(h->f, n2);
  GCALLOC(h->rowptr, ntot + 1);
  GCALLOC(h->col, tot);
  GCALLOC(h->kind, tot);
  GCALLOC(h->stash, (size_t)36 * nc);
  GCALLOC(h->Jc, tot);
  GCALLOC(h->Jv, tot);
  {
    int rcs = mx_alloc_state(h);
    if (rcs) return rcs;
  }
  GCHIP(hipMemcpy(h->coords, m->coords, sizeof(double) * 2 * nv, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->cdofs, cd, sizeof(int32_t) * 6 * (size_t)nc, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->mask, hmask.data(), n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->gbc, hg.data(), sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->phi, p->phi_dofs, sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->f, p->f_dofs, sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (ntot + 1), hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->col, col.data(), sizeof(int32_t) * tot, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->kind, kind.data(), tot, hipMemcpyHostToDevice));
  {
    std::string e1 = pgx_scatter_build(d12.data(), (int64_t)12 * nc, ntot, h->allocs, &h->sc_res);
    if (e1.empty()) e1 = pgx_scatter_build(d36.data(), (int64_t)36 * nc, tot, h->allocs, &h->sc_N);
    if (!e1.empty()) {
      h->err = e1;
      return PGX_ENOMEM;
    }
  }
  // constant blocks, once: park 108 entries per cell, sum per CSR position; the table and the stash are temporary
  GCHIP(hipMemsetAsync(h->Jc, 0, sizeof(double) * tot, h->st));
  {
    std::vector<void*> tmp;
    PgxScatter sc_c;
    std::string e1 = pgx_scatter_build(d108.data(), (int64_t)108 * nc, tot, tmp, &sc_c);
    double* st108 = nullptr;
    hipError_t e = e1.empty() ? hipMalloc((void**)&st108, sizeof(double) * 108 * (size_t)nc) : hipErrorOutOfMemory;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_gc_const, dim3((nc + 127) / 128), dim3(128), 0, h->st, nc, h->cdofs, h->coords, h->Q, st108);
      pgx_scatter_run(h->st, sc_c, st108, 1.0, 0, h->Jc);
      e = hipStreamSynchronize(h->st);
    }
    if (st108) hipFree(st108);
    for (void* q : tmp) hipFree(q);
    if (e != hipSuccess) {
      h->err = std::string("constant Jacobian blocks: ") + (e1.empty() ? hipGetErrorString(e) : e1.c_str());
      return PGX_EHIP;
    }
  }
  return PGX_OK;
}

static GcgArgs gcg_args(const pgx_gc_handle* h) {
  GcgArgs A{};
  A.nc = h->nc, A.n2 = h->n2, A.nv = h->nv, A.NU = h->NU, A.NP = h->NP, A.nq = h->Q.nq;
  A.cells = h->cells3, A.cdu = h->cdofs, A.cdp = h->cdofs_p;
  A.coords = h->coords, A.tNu = h->tNu, A.tdNu = h->tdNu, A.tNp = h->tNp;
  for (int q = 0; q < h->Q.nq; ++q) A.w[q] = h->Q.w[q];
  return A;
}

// general degree: same construction as gc_create_impl with run-time element sizes (see pgx_gc.h: pgx_gc_spaces)
static int gcg_create_impl(pgx_gc_handle* h, const pgx_gc_spaces* sp, const pgx_gc_problem* p, pgx_comm* comm) {
  h->comm = comm;
  h->gen = 1;
  const int NU = sp->nu, NP = sp->np, ND = NU + 2 * NP;
  const int nc = sp->n_cells, n2 = sp->n_u, nv = sp->n_p, nvert = sp->n_vertices;
  const int64_t ntot = (int64_t)n2 + 2 * (int64_t)nv;
  h->NU = NU, h->NP = NP, h->nv = nv, h->nc = nc, h->n2 = n2, h->ntot = ntot;
  h->Q.nq = p->nq;
  for (int q = 0; q < p->nq; ++q) h->Q.X[q] = p->qpts[2 * q], h->Q.Y[q] = p->qpts[2 * q + 1], h->Q.w[q] = p->qwts[q];
  const int32_t *cdu = sp->cell_dofs_u, *cdp = sp->cell_dofs_p;
  for (size_t k = 0; k < (size_t)NU * nc; ++k)
    if (cdu[k] < 0 || cdu[k] >= n2) {
      h->err = "primal cell dof out of range";
      return PGX_EINVAL;
    }
  for (size_t k = 0; k < (size_t)NP * nc; ++k)
    if (cdp[k] < 0 || cdp[k] >= nv) {
      h->err = "latent cell dof out of range";
      return PGX_EINVAL;
    }
  for (size_t k = 0; k < 3 * (size_t)nc; ++k)
    if (sp->cells[k] < 0 || sp->cells[k] >= nvert) {
      h->err = "cell vertex out of range";
      return PGX_EINVAL;
    }
  std::vector<uint8_t> hmask(n2, 0);
  std::vector<double> hg(n2, 0.0);
  for (int k = 0; k < p->n_bc; ++k) {
    const int d = p->bc_dofs[k];
    if (d < 0 || d >= n2) {
      h->err = "bc dof out of range";
      return PGX_EINVAL;
    }
    hmask[d] = 1;
    hg[d] = p->bc_vals ? p->bc_vals[k] : 0.0;
  }
  auto mixed = [&](int c, int32_t* md) {  // NU u dofs, NP psi_x, NP psi_y
    for (int a = 0; a < NU; ++a) md[a] = cdu[(size_t)NU * c + a];
    for (int b = 0; b < NP; ++b) md[NU + b] = n2 + cdp[(size_t)NP * c + b], md[NU + NP + b] = n2 + nv + cdp[(size_t)NP * c + b];
  };
  std::vector<int64_t> dptr(ntot + 1, 0);
  {
    std::vector<int32_t> md(ND);
    for (int c = 0; c < nc; ++c) {
      mixed(c, md.data());
      for (int a = 0; a < ND; ++a) dptr[md[a] + 1]++;
    }
  }
  for (int64_t i = 0; i < ntot; ++i) dptr[i + 1] += dptr[i];
  std::vector<int32_t> dcell(dptr[ntot]);
  {
    std::vector<int64_t> fill(dptr.begin(), dptr.end() - 1);
    std::vector<int32_t> md(ND);
    for (int c = 0; c < nc; ++c) {
      mixed(c, md.data());
      for (int a = 0; a < ND; ++a) dcell[fill[md[a]]++] = c;
    }
  }
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(ntot + 1, 0);
  auto gather_row = [&](int64_t r, std::vector<int32_t>& tmp, std::vector<int32_t>& md) {
    tmp.clear();
    for (int64_t q = dptr[r]; q < dptr[r + 1]; ++q) {
      mixed(dcell[q], md.data());
      tmp.insert(tmp.end(), md.begin(), md.end());
    }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
  };
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp, md(ND);
    for (int64_t r = a; r < b; ++r) {
      gather_row(r, tmp, md);
      rowptr[r + 1] = (int32_t)tmp.size();
    }
  });
  int64_t tot = 0;
  for (int64_t r = 0; r < ntot; ++r) {
    tot += rowptr[r + 1];
    if (tot > 0x7fffffff) {
      h->err = "mixed matrix exceeds int32 nnz";
      return PGX_EINVAL;
    }
    rowptr[r + 1] = (int32_t)tot;
  }
  h->nnz = tot;
  col.resize(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp, md(ND);
    for (int64_t r = a; r < b; ++r) {
      gather_row(r, tmp, md);
      std::copy(tmp.begin(), tmp.end(), col.begin() + rowptr[r]);
    }
  });
  auto find = [&](int32_t r, int32_t c) -> int32_t {
    const int32_t* b = col.data() + rowptr[r];
    const int32_t* e = col.data() + rowptr[r + 1];
    return (int32_t)(std::lower_bound(b, e, c) - col.data());
  };
  std::vector<uint8_t> kind(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r)
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int32_t c = col[k];
        uint8_t t;
        if (r < n2 && c < n2)
          t = (hmask[r] || hmask[c]) ? ((r == c && hmask[r]) ? 3 : 4) : 0;
        else if (r < n2)
          t = hmask[r] ? 4 : 1;
        else if (c < n2)
          t = hmask[c] ? 4 : 1;
        else
          t = 2;
        kind[k] = t;
      }
  });
  const size_t nsc = (size_t)NU * NU + 4 * (size_t)NP * NU, nsn = 4 * (size_t)NP * NP;
  if ((double)(nsc + nsn + ND) * nc * 12.0 > 96e9) {
    h->err = "mesh too large for the one-pass assembly at this degree";
    return PGX_ENOMEM;
  }
  std::vector<int32_t> dC(nsc * nc), dN(nsn * nc), dR((size_t)ND * nc);
  mx_par_for(nc, [&](int64_t a0, int64_t b0) {
    std::vector<int32_t> md(ND);
    for (int64_t c = a0; c < b0; ++c) {
      mixed((int)c, md.data());
      for (int a = 0; a < NU; ++a)
        for (int b = 0; b < NU; ++b) dC[(size_t)(a * NU + b) * nc + (size_t)c] = find(md[a], md[b]);
      for (int b = 0; b < NP; ++b)
        for (int d = 0; d < 2; ++d)
          for (int a = 0; a < NU; ++a) {
            const size_t e = (size_t)NU * NU + ((size_t)(b * 2 + d) * NU + a) * 2;
            const int32_t pr = md[NU + NP * d + b];
            dC[e * nc + (size_t)c] = find(pr, md[a]);
            dC[(e + 1) * nc + (size_t)c] = find(md[a], pr);
          }
      for (int a = 0; a < NP; ++a)
        for (int b = 0; b < NP; ++b)
          for (int cc = 0; cc < 2; ++cc)
            for (int d = 0; d < 2; ++d)
              dN[((size_t)(a * NP + b) * 4 + cc * 2 + d) * nc + (size_t)c] = find(md[NU + NP * cc + a], md[NU + NP * d + b]);
      for (int a = 0; a < ND; ++a) dR[(size_t)a * nc + (size_t)c] = md[a];
    }
  });
  // nested dissection nodes: the primal nodes, then the latent nodes (each carries psi_x and psi_y)
  std::vector<double> xy(2 * ((size_t)n2 + nv));
  std::copy(sp->coords_u, sp->coords_u + 2 * (size_t)n2, xy.begin());
  std::copy(sp->coords_p, sp->coords_p + 2 * (size_t)nv, xy.begin() + 2 * (size_t)n2);
  std::vector<int32_t> nod(ntot);
  for (int i = 0; i < n2; ++i) nod[i] = i;
  for (int v = 0; v < nv; ++v) nod[n2 + v] = nod[(size_t)n2 + nv + v] = n2 + v;
  GCHIP(hipStreamCreate(&h->st));
  pgx_nd_matrix Am{};
  Am.n = ntot;
  Am.rowptr = rowptr.data();
  Am.col = col.data();
  Am.n_nodes = n2 + nv;
  Am.node_of_dof = nod.data();
  Am.dim = 2;
  Am.node_coords = xy.data();
  Am.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) Am.leaf_nodes = atoi(e);
  int rc = comm ? pgx_nd_create_dist(&Am, comm, h->device, (void*)h->st, &h->lu) : pgx_nd_create(&Am, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  pgx_nd_set_symmetric(h->lu, 1);  // (symmetric Newton matrix, as above)
  const int nq = p->nq;
  GCALLOC(h->coords, 2 * (size_t)nvert);
  GCALLOC(h->cells3, 3 * (size_t)nc);
  GCALLOC(h->cdofs, (size_t)NU * nc);
  GCALLOC(h->cdofs_p, (size_t)NP * nc);
  GCALLOC(h->tNu, (size_t)nq * NU);
  GCALLOC(h->tdNu, (size_t)nq * NU * 2);
  GCALLOC(h->tNp, (size_t)nq * NP);
  GCALLOC(h->mask, n2);
  GCALLOC(h->gbc, n2);
  GCALLOC(h->phi, n2);
  GCALLOC(h->f, n2);
  GCALLOC(h->rowptr, ntot + 1);
  GCALLOC(h->col, tot);
  GCALLOC(h->kind, tot);
  GCALLOC(h->stash, std::max(nsn, (size_t)ND) * nc);
  GCALLOC(h->Jc, tot);
  GCALLOC(h->Jv, tot);
  {
    int rcs = mx_alloc_state(h);
    if (rcs) return rcs;
  }
  GCHIP(hipMemcpy(h->coords, sp->coords, sizeof(double) * 2 * nvert, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->cells3, sp->cells, sizeof(int32_t) * 3 * (size_t)nc, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->cdofs, cdu, sizeof(int32_t) * (size_t)NU * nc, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->cdofs_p, cdp, sizeof(int32_t) * (size_t)NP * nc, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->tNu, sp->tab_Nu, sizeof(double) * (size_t)nq * NU, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->tdNu, sp->tab_dNu, sizeof(double) * (size_t)nq * NU * 2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->tNp, sp->tab_Np, sizeof(double) * (size_t)nq * NP, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->mask, hmask.data(), n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->gbc, hg.data(), sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->phi, p->phi_dofs, sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->f, p->f_dofs, sizeof(double) * n2, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (ntot + 1), hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->col, col.data(), sizeof(int32_t) * tot, hipMemcpyHostToDevice));
  GCHIP(hipMemcpy(h->kind, kind.data(), tot, hipMemcpyHostToDevice));
  {
    std::string e1 = pgx_scatter_build(dR.data(), (int64_t)ND * nc, ntot, h->allocs, &h->sc_res);
    if (e1.empty()) e1 = pgx_scatter_build(dN.data(), (int64_t)nsn * nc, tot, h->allocs, &h->sc_N);
    if (!e1.empty()) {
      h->err = e1;
      return PGX_ENOMEM;
    }
  }
  GCHIP(hipMemsetAsync(h->Jc, 0, sizeof(double) * tot, h->st));
  {
    std::vector<void*> tmp;
    PgxScatter sc_c;
    std::string e1 = pgx_scatter_build(dC.data(), (int64_t)nsc * nc, tot, tmp, &sc_c);
    double* stc = nullptr;
    hipError_t e = e1.empty() ? hipMalloc((void**)&stc, sizeof(double) * nsc * nc) : hipErrorOutOfMemory;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_gcg_const, dim3((nc + 63) / 64), dim3(64), 0, h->st, gcg_args(h), stc);
      pgx_scatter_run(h->st, sc_c, stc, 1.0, 0, h->Jc);
      e = hipStreamSynchronize(h->st);
    }
    if (stc) hipFree(stc);
    for (void* q : tmp) hipFree(q);
    if (e != hipSuccess) {
      h->err = std::string("constant Jacobian blocks: ") + (e1.empty() ? hipGetErrorString(e) : e1.c_str());
      return PGX_EHIP;
    }
  }
  return PGX_OK;
}

extern "C" int pgx_gc_create_general(const pgx_gc_spaces* sp, const pgx_gc_problem* p, int device, pgx_gc_handle** out) {
  if (!sp || !p || !out || !sp->coords || !sp->cells || !sp->cell_dofs_u || !sp->cell_dofs_p || !sp->coords_u || !sp->coords_p ||
      !sp->tab_Nu || !sp->tab_dNu || !sp->tab_Np || sp->nu < 3 || sp->nu > GCG_MAXU || sp->np < 1 || sp->np > GCG_MAXP ||
      sp->n_cells <= 0 || sp->n_u <= 0 || sp->n_p <= 0 || !p->qpts || !p->qwts || !p->phi_dofs || !p->f_dofs || p->nq <= 0 ||
      p->nq > GC_MAXQ || (p->n_bc > 0 && !p->bc_dofs)) {
    g_gc_error = "pgx_gc_create_general: bad arguments";
    return PGX_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_gc_error = "pgx_gc_create_general: no usable GPU (there is no CPU fallback)";
    return PGX_ENODEV;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_gc_error = "hipSetDevice failed";
    return PGX_EHIP;
  }
  pgx_gc_handle* h = new pgx_gc_handle();
  h->device = device;
  int rc = gcg_create_impl(h, sp, p, nullptr);
  if (rc) {
    g_gc_error = h->err;
    pgx_gc_destroy(h);
    return rc;
  }
  *out = h;
  return PGX_OK;
}

static int gc_create(const pgx_mesh* m, const pgx_gc_problem* p, pgx_comm* comm, int device, pgx_gc_handle** out) {
  if (!m || !p || !out || !m->coords || !m->cells || !m->cell_dofs || m->n_dofs <= m->n_vertices || !p->qpts || !p->qwts ||
      !p->phi_dofs || !p->f_dofs || p->nq <= 0 || p->nq > GC_MAXQ || (p->n_bc > 0 && !p->bc_dofs)) {
    g_gc_error = "pgx_gc_create: bad arguments";
    return PGX_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_gc_error = "pgx_gc_create: no usable GPU (there is no CPU fallback)";
    return PGX_ENODEV;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_gc_error = "hipSetDevice failed";
    return PGX_EHIP;
  }
  pgx_gc_handle* h = new pgx_gc_handle();
  h->device = device;
  int rc = gc_create_impl(h, m, p, comm);
  if (rc) {
    g_gc_error = h->err;
    pgx_gc_destroy(h);
    return rc;
  }
  *out = h;
  return PGX_OK;
}

extern "C" int pgx_gc_create(const pgx_mesh* m, const pgx_gc_problem* p, int device, pgx_gc_handle** out) {
  return gc_create(m, p, nullptr, device, out);
}
extern "C" int pgx_gc_create_dist(const pgx_mesh* m, const pgx_gc_problem* p, pgx_comm* comm, int device, pgx_gc_handle** out) {
  if (!comm) {
    g_gc_error = "pgx_gc_create_dist: null communicator";
    return PGX_EINVAL;
  }
  return gc_create(m, p, comm, device, out);
}
extern "C" int pgx_gc_lu_stats(const pgx_gc_handle* h, pgx_nd_stats* st) { return h ? pgx_nd_get_stats(h->lu, st) : PGX_EINVAL; }
extern "C" int pgx_gc_lu_is_symmetric(const pgx_gc_handle* h) { return h ? pgx_nd_is_symmetric(h->lu) : 0; }

#define GCNEED(h)                  \
  if (!(h)) return PGX_EINVAL;     \
  if (hipSetDevice((h)->device) != hipSuccess) return PGX_EHIP

extern "C" int pgx_gc_num_dofs(const pgx_gc_handle* h, int64_t* ntot) {
  if (!h || !ntot) return PGX_EINVAL;
  *ntot = h->ntot;
  return PGX_OK;
}
static int gc_in(pgx_gc_handle* h, double* dst, const double* src) { return mx_in(h, dst, src); }
static int gc_out(pgx_gc_handle* h, double* dst, const double* src) { return mx_out(h, dst, src); }
extern "C" int pgx_gc_set_state(pgx_gc_handle* h, const double* x) {
  GCNEED(h);
  return gc_in(h, h->x, x);
}
extern "C" int pgx_gc_get_state(pgx_gc_handle* h, double* x) {
  GCNEED(h);
  return gc_out(h, x, h->x);
}
extern "C" int pgx_gc_set_prev(pgx_gc_handle* h, const double* x) {
  GCNEED(h);
  return gc_in(h, h->xk, x);
}
extern "C" int pgx_gc_get_prev(pgx_gc_handle* h, double* x) {
  GCNEED(h);
  return gc_out(h, x, h->xk);
}
extern "C" int pgx_gc_advance_prev(pgx_gc_handle* h) {
  GCNEED(h);
  GCHIP(hipMemcpyAsync(h->xk, h->x, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));
  GCHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_gc_set_alpha(pgx_gc_handle* h, double a) {
  GCNEED(h);
  if (!(a > 0.0) || !std::isfinite(a)) {
    h->err = "alpha must be positive and finite";
    return PGX_EINVAL;
  }
  h->alpha = a;
  h->jac_valid = false;
  return PGX_OK;
}

void pgx_gc_handle::residual_dev(const double* xin, double* Fout) {
  pgx_gc_handle* h = this;
  GcTimer t(h, 0);
  hipMemsetAsync(Fout, 0, sizeof(double) * h->ntot, h->st);
  if (h->gen) {
    if (h->NU <= GCG_TRIU && h->NP <= GCG_TRIP)
      hipLaunchKernelGGL((k_gcg_residual<GCG_TRIU, GCG_TRIP>), dim3((h->nc + 63) / 64), dim3(64), 0, h->st, gcg_args(h), h->mask, h->gbc,
                         h->phi, h->f, xin, h->xk, h->alpha, h->stash);
    else
      hipLaunchKernelGGL((k_gcg_residual<GCG_MAXU, GCG_MAXP>), dim3((h->nc + 63) / 64), dim3(64), 0, h->st, gcg_args(h), h->mask, h->gbc,
                         h->phi, h->f, xin, h->xk, h->alpha, h->stash);
    pgx_scatter_run(h->st, h->sc_res, h->stash, 1.0, 0, Fout);
    hipLaunchKernelGGL(k_gc_resid_bc, dim3((h->n2 + 255) / 256), dim3(256), 0, h->st, h->n2, h->mask, h->gbc, xin, Fout);
    return;
  }
  hipLaunchKernelGGL(k_gc_residual, dim3((h->nc + 127) / 128), dim3(128), 0, h->st, h->nc, h->n2, h->nv, h->cdofs, h->coords,
                     h->mask, h->gbc, h->phi, h->f, xin, h->xk, h->alpha, h->Q, h->stash);
  pgx_scatter_run(h->st, h->sc_res, h->stash, 1.0, 0, Fout);
  hipLaunchKernelGGL(k_gc_resid_bc, dim3((h->n2 + 255) / 256), dim3(256), 0, h->st, h->n2, h->mask, h->gbc, xin, Fout);
}
void pgx_gc_handle::jacobian_dev(const double* xin) {
  pgx_gc_handle* h = this;
  GcTimer t(h, 1);
  hipLaunchKernelGGL(k_gc_jac_init, dim3((unsigned)((h->nnz + 255) / 256)), dim3(256), 0, h->st, h->nnz, h->kind, h->Jc,
                     h->alpha, h->Jv);
  if (h->gen)
    hipLaunchKernelGGL(k_gcg_jac_N, dim3((h->nc + 63) / 64), dim3(64), 0, h->st, gcg_args(h), h->phi, xin, h->stash);
  else
    hipLaunchKernelGGL(k_gc_jac_N, dim3((h->nc + 127) / 128), dim3(128), 0, h->st, h->nc, h->n2, h->nv, h->cdofs, h->coords,
                       h->phi, xin, h->Q, h->stash);
  pgx_scatter_run(h->st, h->sc_N, h->stash, -1.0, 1, h->Jv);  // the latent block is -N(psi)
  h->jac_valid = true;
}
static void gc_residual_dev(pgx_gc_handle* h, const double* x, double* F) { h->residual_dev(x, F); }
static void gc_jacobian_dev(pgx_gc_handle* h, const double* x) { h->jacobian_dev(x); }
static void gc_spmv_dev(pgx_gc_handle* h, const double* x, double* y) { mx_spmv_dev(h, x, y); }
static int gc_norm(pgx_gc_handle* h, const double* v, double* out) { return mx_norm(h, v, out); }

extern "C" int pgx_gc_residual(pgx_gc_handle* h, const double* x, double* F, double* fnorm) {
  GCNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = gc_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  gc_residual_dev(h, xd, h->F);
  if (fnorm) {
    int rc = gc_norm(h, h->F, fnorm);
    if (rc) return rc;
  }
  if (F) return gc_out(h, F, h->F);
  GCHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}

extern "C" int pgx_gc_jacobian_fill(pgx_gc_handle* h, const double* x) {
  GCNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = gc_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  gc_jacobian_dev(h, xd);
  GCHIP(hipStreamSynchronize(h->st));
  GCHIP(hipGetLastError());
  return PGX_OK;
}

extern "C" int pgx_gc_csr_export(pgx_gc_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col,
                                 double* vals) {
  GCNEED(h);
  if (nrows) *nrows = h->ntot;
  if (nnz) *nnz = h->nnz;
  if (rowptr) std::copy(h->h_rowptr.begin(), h->h_rowptr.end(), rowptr);
  if (col) std::copy(h->h_col.begin(), h->h_col.end(), col);
  if (vals) {
    if (!h->jac_valid) {
      h->err = "pgx_gc_csr_export: no Jacobian has been filled";
      return PGX_ESTATE;
    }
    GCHIP(hipMemcpy(vals, h->Jv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  }
  return PGX_OK;
}

extern "C" int pgx_gc_spmv(pgx_gc_handle* h, const double* x, double* y) {
  GCNEED(h);
  if (!x || !y) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_gc_spmv: no Jacobian has been filled";
    return PGX_ESTATE;
  }
  int rc = gc_in(h, h->r, x);
  if (rc) return rc;
  gc_spmv_dev(h, h->r, h->z);
  return gc_out(h, y, h->z);
}

extern "C" int pgx_gc_l2_increment(pgx_gc_handle* h, double* out) {
  GCNEED(h);
  if (!out) return PGX_EINVAL;
  if (h->gen)
    hipLaunchKernelGGL(k_gcg_l2, dim3(GC_RED), dim3(256), 0, h->st, gcg_args(h), h->x, h->xk, h->partials);
  else
    hipLaunchKernelGGL(k_gc_l2, dim3(GC_RED), dim3(256), 0, h->st, h->nc, h->cdofs, h->coords, h->x, h->xk, h->Q, h->partials);
  hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, GC_RED, h->partials, h->d_out);
  {
    const int rcs = mx_sync_scalar(h);  // distributed handles: the loop's stopping test must agree on every rank
    if (rcs) return rcs;
  }
  GCHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  GCHIP(hipStreamSynchronize(h->st));
  *out = std::sqrt(std::max(h->h_out[0], 0.0));
  return PGX_OK;
}

extern "C" int pgx_gc_profile(pgx_gc_handle* h, int enable, double ms[6]) {
  GCNEED(h);
  pgx_nd_timing(h->lu, enable, nullptr, nullptr);
  return mx_profile(h, enable, ms);
}

extern "C" int pgx_gc_newton_solve(pgx_gc_handle* h, const pgx_snes_opts* opts, int* reason, int* its_out, int* lin_out) {
  GCNEED(h);
  if (!opts) return PGX_EINVAL;
  return opts->linesearch == 1 ? mx_newton_solve_bt(h, opts, reason, its_out, lin_out)
                               : mx_newton_solve(h, opts, reason, its_out, lin_out);
}
