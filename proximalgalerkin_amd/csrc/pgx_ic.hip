// pgx_ic.hip - example 08 (intersecting constraints: u, psi0, psi in P1 on an interval) behind the C ABI of include/pgx_ic.h.
// Reference: examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py (:13-23 spaces, :30-45 data, :47-58
// residual, :60-63 BCs, :66-79 solver, :112-175 loop).  x = [u | psi0 | psi]; the 3 x 3 block Jacobian lives in one mixed CSR
// array on the chain's pattern (row (field, i): columns (0..2, i-1..i+1)).  Both assembly kernels are VERTEX-parallel: a thread
// owns the three rows of its vertex and walks its (at most two) cells in a fixed order - no scatter, no atomics, bitwise
// reproducible.  The problem is tiny (3 006 unknowns in the reference's configuration): the point of this file is that the
// composition of example 01's and example 06's latent rows runs through the same Newton / sparse-LU machinery as the others.
#include <cstring>

#include "../../include/pgx_ic.h"
#include "pgx_mixed.h"

#define IC_MAXQ 16
struct IcQuad {
  double N[IC_MAXQ][2], w[IC_MAXQ];
  int nq;
};

static thread_local std::string g_ic_error;

struct pgx_ic_handle : MixedBase {
  int nv = 0, nc = 0;
  IcQuad Q{};
  double alpha = 1.0, c = 0.0;
  double *xc = nullptr, *phi0_q = nullptr, *phi_q = nullptr;
  uint8_t* mask = nullptr;
  void residual_dev(const double* xin, double* Fout) override;
  void jacobian_dev(const double* xin) override;
};

extern "C" const char* pgx_ic_last_error(const pgx_ic_handle* h) { return h ? h->err.c_str() : g_ic_error.c_str(); }

// R_u, R_psi0, R_psi of vertex i: contributions of cell i - 1 (local index 1), then of cell i (local index 0)
__global__ __launch_bounds__(128) void k_ic_residual(int nv, const double* __restrict__ xc, const uint8_t* __restrict__ mask,
                                                     const double* __restrict__ x, const double* __restrict__ xk,
                                                     const double* __restrict__ phi0_q, const double* __restrict__ phi_q,
                                                     double alpha, double c, IcQuad Q, double* __restrict__ F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  double Ru = 0.0, R0 = 0.0, Rp = 0.0;
  for (int side = 0; side < 2; ++side) {
    const int e = i - 1 + side, a = 1 - side;
    if (e < 0 || e >= nv - 1) continue;
    const double h = xc[e + 1] - xc[e];
    const double dN[2] = {-1.0 / h, 1.0 / h};
    double u[2], p0[2], p[2], p0k[2], pk[2];
    for (int b = 0; b < 2; ++b) {
      const int v = e + b;
      u[b] = mask[v] ? 0.0 : x[v];  // the residual is assembled with the boundary values in place
      p0[b] = x[nv + v], p[b] = x[2 * nv + v];
      p0k[b] = xk[nv + v], pk[b] = xk[2 * nv + v];
    }
    const double du = u[0] * dN[0] + u[1] * dN[1];
    Ru += alpha * h * du * dN[a];
    for (int q = 0; q < Q.nq; ++q) {
      const double* N = Q.N[q];
      const double wd = Q.w[q] * h;
      const double uq = u[0] * N[0] + u[1] * N[1];
      const double p0q = p0[0] * N[0] + p0[1] * N[1], pq = p[0] * N[0] + p[1] * N[1];
      const double p0kq = p0k[0] * N[0] + p0k[1] * N[1], pkq = pk[0] * N[0] + pk[1] * N[1];
      Ru += wd * ((alpha * c + p0q - p0kq) * N[a] + (pq - pkq) * dN[a]);
      R0 += wd * (uq - exp(p0q) - phi0_q[(size_t)e * Q.nq + q]) * N[a];
      Rp += wd * (du - phi_q[(size_t)e * Q.nq + q] * pq / sqrt(1.0 + pq * pq)) * N[a];
    }
  }
  F[i] = mask[i] ? x[i] : Ru;
  F[nv + i] = R0;
  F[2 * nv + i] = Rp;
}

// the 27 (18 at the ends) entries of vertex i's three rows; CSR position of (row field fr, column field fc, neighbour k):
// rowptr[fr * nv + i] + fc * len + k, len = number of neighbours incl. i, k counted from max(i - 1, 0)
__global__ __launch_bounds__(128) void k_ic_jacobian(int nv, const double* __restrict__ xc, const uint8_t* __restrict__ mask,
                                                     const double* __restrict__ x, const double* __restrict__ phi_q, double alpha,
                                                     IcQuad Q, const int32_t* __restrict__ rowptr, double* __restrict__ Jv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  const int jlo = i > 0 ? i - 1 : 0, jhi = i < nv - 1 ? i + 1 : nv - 1, len = jhi - jlo + 1;
  double acc[3][3][3];
  for (int r = 0; r < 3; ++r)
    for (int s = 0; s < 3; ++s)
      for (int k = 0; k < 3; ++k) acc[r][s][k] = 0.0;
  for (int side = 0; side < 2; ++side) {
    const int e = i - 1 + side, a = 1 - side;
    if (e < 0 || e >= nv - 1) continue;
    const double h = xc[e + 1] - xc[e];
    const double dN[2] = {-1.0 / h, 1.0 / h};
    const double p00 = x[nv + e], p01 = x[nv + e + 1], p0 = x[2 * nv + e], p1 = x[2 * nv + e + 1];
    double M[2] = {0, 0}, D0[2] = {0, 0}, D[2] = {0, 0}, Na = 0.0;  // row a against columns b = 0, 1
    for (int q = 0; q < Q.nq; ++q) {
      const double* N = Q.N[q];
      const double wd = Q.w[q] * h;
      const double e0 = exp(p00 * N[0] + p01 * N[1]);
      const double pq = p0 * N[0] + p1 * N[1];
      const double t = 1.0 + pq * pq;
      const double dh = phi_q[(size_t)e * Q.nq + q] / (t * sqrt(t));
      Na += wd * N[a];
      for (int b = 0; b < 2; ++b) {
        const double nn = wd * N[a] * N[b];
        M[b] += nn, D0[b] += nn * e0, D[b] += nn * dh;
      }
    }
    for (int b = 0; b < 2; ++b) {
      const int k = e + b - jlo;
      double Nb = 0.0;
      for (int q = 0; q < Q.nq; ++q) Nb += Q.w[q] * h * Q.N[q][b];
      acc[0][0][k] += alpha * h * dN[a] * dN[b];
      acc[0][1][k] += M[b];
      acc[0][2][k] += Nb * dN[a];  // (psi_b, v_a')
      acc[1][0][k] += M[b];
      acc[1][1][k] -= D0[b];
      acc[2][0][k] += Na * dN[b];  // (u_b', w_a)
      acc[2][2][k] -= D[b];
    }
  }
  for (int fr = 0; fr < 3; ++fr) {
    const int64_t base = rowptr[(size_t)fr * nv + i];
    for (int fc = 0; fc < 3; ++fc)
      for (int k = 0; k < len; ++k) {
        const int j = jlo + k;
        double v = acc[fr][fc][k];
        if (fr == 0 && mask[i]) v = (fc == 0 && j == i) ? 1.0 : 0.0;  // Dirichlet row
        else if (fc == 0 && mask[j]) v = 0.0;                          // Dirichlet column
        Jv[base + (int64_t)fc * len + k] = v;
      }
  }
}

// sum over cells of (d, d), d = u - u_iter: per-block partials
__global__ __launch_bounds__(256) void k_ic_l2(int nc, const double* __restrict__ xc, const double* __restrict__ x,
                                               const double* __restrict__ xk, IcQuad Q, double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nc; e += MX_RED * 256) {
    const double h = xc[e + 1] - xc[e], d0 = x[e] - xk[e], d1 = x[e + 1] - xk[e + 1];
    for (int q = 0; q < Q.nq; ++q) {
      const double v = d0 * Q.N[q][0] + d1 * Q.N[q][1];
      s += Q.w[q] * h * v * v;
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
extern "C" void pgx_ic_destroy(pgx_ic_handle* h) {
  if (!h) return;
  mx_release(h);
  delete h;
}

void pgx_ic_handle::residual_dev(const double* xin, double* Fout) {
  pgx_ic_handle* h = this;
  MxTimer t(h, 0);
  hipLaunchKernelGGL(k_ic_residual, dim3((h->nv + 127) / 128), dim3(128), 0, h->st, h->nv, h->xc, h->mask, xin, h->xk, h->phi0_q,
                     h->phi_q, h->alpha, h->c, h->Q, Fout);
}
void pgx_ic_handle::jacobian_dev(const double* xin) {
  pgx_ic_handle* h = this;
  MxTimer t(h, 1);
  hipLaunchKernelGGL(k_ic_jacobian, dim3((h->nv + 127) / 128), dim3(128), 0, h->st, h->nv, h->xc, h->mask, xin, h->phi_q,
                     h->alpha, h->Q, h->rowptr, h->Jv);
  h->jac_valid = true;
}

static int ic_create_impl(pgx_ic_handle* h, const pgx_ic_problem* p) {
  const int nv = p->n_vertices, nc = nv - 1;
  const int64_t ntot = 3 * (int64_t)nv;
  h->nv = nv, h->nc = nc, h->ntot = ntot, h->c = p->c;
  h->Q.nq = p->nq;
  for (int q = 0; q < p->nq; ++q) {
    if (!(p->qpts[q] > 0.0 && p->qpts[q] < 1.0)) {
      h->err = "quadrature points must lie in (0, 1)";
      return PGX_EINVAL;
    }
    h->Q.N[q][0] = 1.0 - p->qpts[q], h->Q.N[q][1] = p->qpts[q], h->Q.w[q] = p->qwts[q];
  }
  for (int v = 0; v + 1 < nv; ++v)
    if (!(p->x[v + 1] > p->x[v])) {
      h->err = "vertex coordinates must be strictly increasing";
      return PGX_EINVAL;
    }
  std::vector<uint8_t> hmask(nv, 0);
  for (int k = 0; k < p->n_bc; ++k) {
    if (p->bc_dofs[k] < 0 || p->bc_dofs[k] >= nv) {
      h->err = "bc dof out of range";
      return PGX_EINVAL;
    }
    hmask[p->bc_dofs[k]] = 1;
  }
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(ntot + 1, 0);
  auto lo = [&](int i) { return i > 0 ? i - 1 : 0; };
  auto hi = [&](int i) { return i < nv - 1 ? i + 1 : nv - 1; };
  for (int fr = 0; fr < 3; ++fr)
    for (int i = 0; i < nv; ++i) rowptr[(size_t)fr * nv + i + 1] = 3 * (hi(i) - lo(i) + 1);
  for (int64_t r = 0; r < ntot; ++r) rowptr[r + 1] += rowptr[r];
  const int64_t tot = rowptr[ntot];
  h->nnz = tot;
  col.resize(tot);
  for (int fr = 0; fr < 3; ++fr)
    for (int i = 0; i < nv; ++i) {
      const int len = hi(i) - lo(i) + 1;
      for (int fc = 0; fc < 3; ++fc)
        for (int k = 0; k < len; ++k) col[rowptr[(size_t)fr * nv + i] + (size_t)fc * len + k] = fc * nv + lo(i) + k;
    }
  std::vector<int32_t> nod(ntot);
  std::vector<double> xy(2 * (size_t)nv, 0.0);  // the chain embedded in the plane for the dissection: (x, 0)
  for (int v = 0; v < nv; ++v) {
    nod[v] = nod[(size_t)nv + v] = nod[2 * (size_t)nv + v] = v;
    xy[2 * (size_t)v] = p->x[v];
  }
  MXHIP(hipStreamCreate(&h->st));
  pgx_nd_matrix A{};
  A.n = ntot;
  A.rowptr = rowptr.data();
  A.col = col.data();
  A.n_nodes = nv;
  A.node_of_dof = nod.data();
  A.dim = 2;
  A.node_coords = xy.data();
  A.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) A.leaf_nodes = atoi(e);
  int rc = pgx_nd_create(&A, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  const size_t nqc = (size_t)nc * p->nq;
  MXALLOC(h->xc, nv);
  MXALLOC(h->mask, nv);
  MXALLOC(h->phi0_q, nqc);
  MXALLOC(h->phi_q, nqc);
  MXALLOC(h->rowptr, ntot + 1);
  MXALLOC(h->col, tot);
  MXALLOC(h->Jv, tot);
  if ((rc = mx_alloc_state(h))) return rc;
  MXHIP(hipMemcpy(h->xc, p->x, sizeof(double) * nv, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->mask, hmask.data(), nv, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->phi0_q, p->phi0_q, sizeof(double) * nqc, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->phi_q, p->phi_q, sizeof(double) * nqc, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (ntot + 1), hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->col, col.data(), sizeof(int32_t) * tot, hipMemcpyHostToDevice));
  return PGX_OK;
}

extern "C" int pgx_ic_create(const pgx_ic_problem* p, int device, pgx_ic_handle** out) {
  if (!p || !out || !p->x || p->n_vertices < 2 || !p->qpts || !p->qwts || p->nq <= 0 || p->nq > IC_MAXQ || !p->phi0_q ||
      !p->phi_q || (p->n_bc > 0 && !p->bc_dofs)) {
    g_ic_error = "pgx_ic_create: bad arguments";
    return PGX_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_ic_error = "pgx_ic_create: no usable GPU (there is no CPU fallback)";
    return PGX_ENODEV;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_ic_error = "hipSetDevice failed";
    return PGX_EHIP;
  }
  pgx_ic_handle* h = new pgx_ic_handle();
  h->device = device;
  int rc = ic_create_impl(h, p);
  if (rc) {
    g_ic_error = h->err;
    pgx_ic_destroy(h);
    return rc;
  }
  *out = h;
  return PGX_OK;
}

#define ICNEED(h)              \
  if (!(h)) return PGX_EINVAL; \
  if (hipSetDevice((h)->device) != hipSuccess) return PGX_EHIP

extern "C" int pgx_ic_num_dofs(const pgx_ic_handle* h, int64_t* ntot) {
  if (!h || !ntot) return PGX_EINVAL;
  *ntot = h->ntot;
  return PGX_OK;
}
extern "C" int pgx_ic_set_state(pgx_ic_handle* h, const double* x) {
  ICNEED(h);
  return mx_in(h, h->x, x);
}
extern "C" int pgx_ic_get_state(pgx_ic_handle* h, double* x) {
  ICNEED(h);
  return mx_out(h, x, h->x);
}
extern "C" int pgx_ic_set_prev(pgx_ic_handle* h, const double* x) {
  ICNEED(h);
  return mx_in(h, h->xk, x);
}
extern "C" int pgx_ic_get_prev(pgx_ic_handle* h, double* x) {
  ICNEED(h);
  return mx_out(h, x, h->xk);
}
extern "C" int pgx_ic_advance_prev(pgx_ic_handle* h) {
  ICNEED(h);
  MXHIP(hipMemcpyAsync(h->xk, h->x, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_ic_set_alpha(pgx_ic_handle* h, double a) {
  ICNEED(h);
  if (!(a > 0.0) || !std::isfinite(a)) {
    h->err = "alpha must be positive and finite";
    return PGX_EINVAL;
  }
  h->alpha = a;
  h->jac_valid = false;
  return PGX_OK;
}
extern "C" int pgx_ic_set_phi(pgx_ic_handle* h, const double* phi_q) {
  ICNEED(h);
  h->jac_valid = false;
  return mx_in(h, h->phi_q, phi_q, (int64_t)h->nc * h->Q.nq);
}
extern "C" int pgx_ic_residual(pgx_ic_handle* h, const double* x, double* F, double* fnorm) {
  ICNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = mx_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  h->residual_dev(xd, h->F);
  if (fnorm) {
    int rc = mx_norm(h, h->F, fnorm);
    if (rc) return rc;
  }
  if (F) return mx_out(h, F, h->F);
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_ic_jacobian_fill(pgx_ic_handle* h, const double* x) {
  ICNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = mx_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  h->jacobian_dev(xd);
  MXHIP(hipStreamSynchronize(h->st));
  MXHIP(hipGetLastError());
  return PGX_OK;
}
extern "C" int pgx_ic_csr_export(pgx_ic_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* vals) {
  ICNEED(h);
  if (nrows) *nrows = h->ntot;
  if (nnz) *nnz = h->nnz;
  if (rowptr) std::copy(h->h_rowptr.begin(), h->h_rowptr.end(), rowptr);
  if (col) std::copy(h->h_col.begin(), h->h_col.end(), col);
  if (vals) {
    if (!h->jac_valid) {
      h->err = "pgx_ic_csr_export: no Jacobian has been filled";
      return PGX_ESTATE;
    }
    MXHIP(hipMemcpy(vals, h->Jv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  }
  return PGX_OK;
}
extern "C" int pgx_ic_spmv(pgx_ic_handle* h, const double* x, double* y) {
  ICNEED(h);
  if (!x || !y) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_ic_spmv: no Jacobian has been filled";
    return PGX_ESTATE;
  }
  int rc = mx_in(h, h->r, x);
  if (rc) return rc;
  mx_spmv_dev(h, h->r, h->z);
  return mx_out(h, y, h->z);
}
extern "C" int pgx_ic_newton_solve(pgx_ic_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its) {
  ICNEED(h);
  if (!opts) return PGX_EINVAL;
  switch (opts->linesearch) {
    case 2: return mx_newton_solve_l2(h, opts, reason, its, lin_its);
    case 1: return mx_newton_solve_bt(h, opts, reason, its, lin_its);
    default: return mx_newton_solve(h, opts, reason, its, lin_its);
  }
}
extern "C" int pgx_ic_l2_increment(pgx_ic_handle* h, double* out) {
  ICNEED(h);
  if (!out) return PGX_EINVAL;
  hipLaunchKernelGGL(k_ic_l2, dim3(MX_RED), dim3(256), 0, h->st, h->nc, h->xc, h->x, h->xk, h->Q, h->partials);
  hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
  MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  *out = std::sqrt(std::max(h->h_out[0], 0.0));
  return PGX_OK;
}
extern "C" int pgx_ic_profile(pgx_ic_handle* h, int enable, double ms[6]) {
  ICNEED(h);
  pgx_nd_timing(h->lu, enable, nullptr, nullptr);
  return mx_profile(h, enable, ms);
}
