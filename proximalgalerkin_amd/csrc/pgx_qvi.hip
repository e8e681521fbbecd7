// pgx_qvi.hip - example 05 (thermoforming QVI: u, T, psi in P1) behind the C ABI of include/pgx_qvi.h.
// Reference: examples/05_obstacle_type_qvi/thermoforming_dolfinx.py (:28-33 spaces, :36-59 data, :62-71 residual and the
// modified Jacobian, :100-116 solver, :117-158 loop).  x = [u | T | psi]; the 3 x 3 block Jacobian lives in one mixed CSR
// array (9 blocks on the scalar P1 pattern, the two structurally zero ones included); constant blocks (K, M, K + beta M,
// -M_xi) are assembled once, every Newton step re-assembles the two psi-dependent blocks (T,psi) and (psi,psi).
#include <cstring>

#include "../../include/pgx_qvi.h"
#include "pgx_mixed.h"
#include "pgx_scatter.h"

#define QV_MAXQ 16
struct QvQuad {
  double N[QV_MAXQ][3], w[QV_MAXQ];
  int nq;
};

static thread_local std::string g_qvi_error;

struct pgx_qvi_handle : MixedBase {
  int nv = 0, nc = 0;
  QvQuad Q{};
  double alpha = 1.0, beta = 1.0, f = 25.0, knee = 0.01, eps_mod = 1e-10;
  double* coords = nullptr;
  int32_t* cells = nullptr;
  // deterministic assembly (pgx_scatter.h): element kernels park [slot * nc + cell]; one thread per destination sums
  PgxScatter sc_res, sc_psi;  // residual: 9 slots per cell -> dofs; psi-dependent blocks: 18 slots per cell -> CSR positions
  double* stash = nullptr;    // [18 * nc]
  uint8_t *mask = nullptr, *kind = nullptr;
  double* Jc = nullptr;
  void residual_dev(const double* xin, double* Fout) override;
  void jacobian_dev(const double* xin) override;
};

extern "C" const char* pgx_qvi_last_error(const pgx_qvi_handle* h) { return h ? h->err.c_str() : g_qvi_error.c_str(); }

struct QvGeom {
  double G[3][2];  // physical P1 gradients
  double adet;
  double X[3][2];
};
__device__ inline QvGeom qv_geom(const double* __restrict__ coords, const int32_t* __restrict__ cv) {
  QvGeom g;
  for (int a = 0; a < 3; ++a) g.X[a][0] = coords[2 * (size_t)cv[a]], g.X[a][1] = coords[2 * (size_t)cv[a] + 1];
  const double j00 = g.X[1][0] - g.X[0][0], j10 = g.X[1][1] - g.X[0][1];
  const double j01 = g.X[2][0] - g.X[0][0], j11 = g.X[2][1] - g.X[0][1];
  const double det = j00 * j11 - j01 * j10;
  const double i00 = j11 / det, i01 = -j01 / det, i10 = -j10 / det, i11 = j00 / det;
  // G[a][d] = sum_k gref[a][k] inv[k][d], gref = [[-1,-1],[1,0],[0,1]]
  g.G[1][0] = i00, g.G[1][1] = i01;
  g.G[2][0] = i10, g.G[2][1] = i11;
  g.G[0][0] = -(i00 + i10), g.G[0][1] = -(i01 + i11);
  g.adet = fabs(det);
  return g;
}
__device__ inline void qv_space(const QvGeom& g, const double N[3], double* phi0, double* xi) {
  const double x = N[0] * g.X[0][0] + N[1] * g.X[1][0] + N[2] * g.X[2][0];
  const double y = N[0] * g.X[0][1] + N[1] * g.X[1][1] + N[2] * g.X[2][1];
  *phi0 = 1.0 - 2.0 * fmax(fabs(x - 0.5), fabs(y - 0.5));          // thermoforming_dolfinx.py:58
  *xi = sin(3.14159265358979323846 * x) * sin(3.14159265358979323846 * y);  // :59
}

__global__ __launch_bounds__(128) void k_qv_residual(int nc, int nv, const int32_t* __restrict__ cells,
                                                     const double* __restrict__ coords, const uint8_t* __restrict__ mask,
                                                     const double* __restrict__ x, const double* __restrict__ xk, double alpha,
                                                     double beta, double f, double knee, QvQuad Q, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + 3 * (size_t)c;
  const QvGeom g = qv_geom(coords, cv);
  double u[3], T[3], p[3], pk[3];
  for (int a = 0; a < 3; ++a) {
    const int v = cv[a];
    u[a] = mask[v] ? 0.0 : x[v];
    T[a] = x[nv + v];
    p[a] = x[2 * nv + v];
    pk[a] = xk[2 * nv + v];
  }
  const double area = 0.5 * g.adet;
  double gu[2] = {0, 0}, gT[2] = {0, 0};
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 2; ++d) gu[d] += u[a] * g.G[a][d], gT[d] += T[a] * g.G[a][d];
  double Ru[3], RT[3], Rp[3] = {0, 0, 0};
  for (int a = 0; a < 3; ++a) {
    Ru[a] = alpha * area * (gu[0] * g.G[a][0] + gu[1] * g.G[a][1]);
    RT[a] = area * (gT[0] * g.G[a][0] + gT[1] * g.G[a][1]);
  }
  for (int q = 0; q < Q.nq; ++q) {
    const double* N = Q.N[q];
    const double wd = Q.w[q] * g.adet;
    const double uq = u[0] * N[0] + u[1] * N[1] + u[2] * N[2], Tq = T[0] * N[0] + T[1] * N[1] + T[2] * N[2];
    const double pq = p[0] * N[0] + p[1] * N[1] + p[2] * N[2], pkq = pk[0] * N[0] + pk[1] * N[1] + pk[2] * N[2];
    const double s = exp(-pq);
    const double gv = s < knee ? 1.0 - s / knee : 0.0;
    double phi0, xi;
    qv_space(g, N, &phi0, &xi);
    const double cu = wd * (pq - pkq - alpha * f), cT = wd * (beta * Tq - gv), cp = wd * (uq + s - phi0 - xi * Tq);
    for (int a = 0; a < 3; ++a) Ru[a] += cu * N[a], RT[a] += cT * N[a], Rp[a] += cp * N[a];
  }
  for (int a = 0; a < 3; ++a) {  // parked slot-major; pgx_scatter sums per dof in a fixed order
    stash[(size_t)a * nc + c] = Ru[a];
    stash[(size_t)(3 + a) * nc + c] = RT[a];
    stash[(size_t)(6 + a) * nc + c] = Rp[a];
  }
}
__global__ void k_qv_resid_bc(int nv, const uint8_t* __restrict__ mask, const double* __restrict__ x, double* __restrict__ F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv && mask[i]) F[i] = x[i];
}

// constant blocks, once: five 3x3 blocks per cell parked at stash[(blk * 9 + a * 3 + b) * nc + cell], blk = (u,u), (u,psi),
// (T,T), (psi,u), (psi,T)
__global__ __launch_bounds__(128) void k_qv_const(int nc, const int32_t* __restrict__ cells, const double* __restrict__ coords,
                                                  double beta, QvQuad Q, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + 3 * (size_t)c;
  const QvGeom g = qv_geom(coords, cv);
  double Ke[3][3], Me[3][3], Mx[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      Ke[a][b] = 0.5 * g.adet * (g.G[a][0] * g.G[b][0] + g.G[a][1] * g.G[b][1]);
      Me[a][b] = Mx[a][b] = 0.0;
    }
  for (int q = 0; q < Q.nq; ++q) {
    const double* N = Q.N[q];
    const double wd = Q.w[q] * g.adet;
    double phi0, xi;
    qv_space(g, N, &phi0, &xi);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Me[a][b] += wd * N[a] * N[b], Mx[a][b] += wd * xi * N[a] * N[b];
  }
  auto park = [&](int blk, int a, int b, double v) { stash[(size_t)(blk * 9 + a * 3 + b) * nc + c] = v; };
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      park(0, a, b, Ke[a][b]);                    // (u,u): K, scaled by alpha per step
      park(1, a, b, Me[a][b]);                    // (u,psi): M
      park(2, a, b, Ke[a][b] + beta * Me[a][b]);  // (T,T)
      park(3, a, b, Me[a][b]);                    // (psi,u): M
      park(4, a, b, -Mx[a][b]);                   // (psi,T): -M_xi
    }
}

// kind: 0 = (u,u) slot scaled by alpha, 1 = constant slot, 2 = psi-dependent slot (k_qv_jac_psi), 3 = BC diagonal, 4 = zero
__global__ void k_qv_jac_init(int64_t nnz, const uint8_t* __restrict__ kind, const double* __restrict__ Jc, double alpha,
                              double* __restrict__ Jv) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int t = kind[k];
  Jv[k] = t == 0 ? alpha * Jc[k] : t == 1 ? Jc[k] : t == 3 ? 1.0 : 0.0;
}

// (T,psi): C_e = int g'(s) s N_a N_b, g' = -1/knee on (0, knee);  (psi,psi): -int s N_a N_b - (eps/alpha) K_e   (:69-71)
__global__ __launch_bounds__(128) void k_qv_jac_psi(int nc, int nv, const int32_t* __restrict__ cells,
                                                    const double* __restrict__ coords, const double* __restrict__ x,
                                                    double knee, double eps_over_alpha, QvQuad Q,
                                                    double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + 3 * (size_t)c;
  const QvGeom g = qv_geom(coords, cv);
  const double p0 = x[2 * nv + cv[0]], p1 = x[2 * nv + cv[1]], p2 = x[2 * nv + cv[2]];
  double Ce[3][3], De[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      Ce[a][b] = 0.0;
      De[a][b] = -eps_over_alpha * 0.5 * g.adet * (g.G[a][0] * g.G[b][0] + g.G[a][1] * g.G[b][1]);
    }
  for (int q = 0; q < Q.nq; ++q) {
    const double* N = Q.N[q];
    const double s = exp(-(p0 * N[0] + p1 * N[1] + p2 * N[2]));
    const double ws = Q.w[q] * g.adet * s;
    const double wc = s < knee ? -ws / knee : 0.0;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        Ce[a][b] += wc * N[a] * N[b];
        De[a][b] -= ws * N[a] * N[b];
      }
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      stash[(size_t)(a * 3 + b) * nc + c] = Ce[a][b];
      stash[(size_t)(9 + a * 3 + b) * nc + c] = De[a][b];
    }
}

// sum over cells of (d, d) + (grad d, grad d), d = u - u_prev: per-block partials
__global__ __launch_bounds__(256) void k_qv_h1(int nc, const int32_t* __restrict__ cells, const double* __restrict__ coords,
                                               const double* __restrict__ x, const double* __restrict__ xk, QvQuad Q,
                                               double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < nc; c += MX_RED * 256) {
    const int32_t* cv = cells + 3 * (size_t)c;
    const QvGeom g = qv_geom(coords, cv);
    double d[3], gd[2] = {0, 0};
    for (int a = 0; a < 3; ++a) {
      d[a] = x[cv[a]] - xk[cv[a]];
      gd[0] += d[a] * g.G[a][0];
      gd[1] += d[a] * g.G[a][1];
    }
    s += 0.5 * g.adet * (gd[0] * gd[0] + gd[1] * gd[1]);
    for (int q = 0; q < Q.nq; ++q) {
      const double v = d[0] * Q.N[q][0] + d[1] * Q.N[q][1] + d[2] * Q.N[q][2];
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
extern "C" void pgx_qvi_destroy(pgx_qvi_handle* h) {
  if (!h) return;
  mx_release(h);
  delete h;
}

void pgx_qvi_handle::residual_dev(const double* xin, double* Fout) {
  pgx_qvi_handle* h = this;
  MxTimer t(h, 0);
  hipMemsetAsync(Fout, 0, sizeof(double) * h->ntot, h->st);
  hipLaunchKernelGGL(k_qv_residual, dim3((h->nc + 127) / 128), dim3(128), 0, h->st, h->nc, h->nv, h->cells, h->coords, h->mask,
                     xin, h->xk, h->alpha, h->beta, h->f, h->knee, h->Q, h->stash);
  pgx_scatter_run(h->st, h->sc_res, h->stash, 1.0, 0, Fout);
  hipLaunchKernelGGL(k_qv_resid_bc, dim3((h->nv + 255) / 256), dim3(256), 0, h->st, h->nv, h->mask, xin, Fout);
}
void pgx_qvi_handle::jacobian_dev(const double* xin) {
  pgx_qvi_handle* h = this;
  MxTimer t(h, 1);
  hipLaunchKernelGGL(k_qv_jac_init, dim3((unsigned)((h->nnz + 255) / 256)), dim3(256), 0, h->st, h->nnz, h->kind, h->Jc,
                     h->alpha, h->Jv);
  hipLaunchKernelGGL(k_qv_jac_psi, dim3((h->nc + 127) / 128), dim3(128), 0, h->st, h->nc, h->nv, h->cells, h->coords, xin,
                     h->knee, h->eps_mod / h->alpha, h->Q, h->stash);
  pgx_scatter_run(h->st, h->sc_psi, h->stash, 1.0, 1, h->Jv);
  h->jac_valid = true;
}

static int qvi_create_impl(pgx_qvi_handle* h, const pgx_mesh* m, const pgx_qvi_problem* p) {
  const int nv = m->n_vertices, nc = m->n_cells;
  const int64_t ntot = 3 * (int64_t)nv;
  h->nv = nv, h->nc = nc, h->ntot = ntot;
  h->beta = p->beta, h->f = p->f, h->knee = p->knee, h->eps_mod = p->eps_mod;
  h->Q.nq = p->nq;
  for (int q = 0; q < p->nq; ++q) {
    const double X = p->qpts[2 * q], Y = p->qpts[2 * q + 1];
    h->Q.N[q][0] = 1.0 - X - Y, h->Q.N[q][1] = X, h->Q.N[q][2] = Y, h->Q.w[q] = p->qwts[q];
  }
  for (size_t k = 0; k < 3 * (size_t)nc; ++k)
    if (m->cells[k] < 0 || m->cells[k] >= nv) {
      h->err = "cell vertex out of range";
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
  // scalar P1 pattern (vertex adjacency incl. self), then 3 x 3 blocks
  std::vector<int64_t> vptr(nv + 1, 0);
  for (size_t k = 0; k < 3 * (size_t)nc; ++k) vptr[m->cells[k] + 1]++;
  for (int v = 0; v < nv; ++v) vptr[v + 1] += vptr[v];
  std::vector<int32_t> vcell(vptr[nv]);
  {
    std::vector<int64_t> fill(vptr.begin(), vptr.end() - 1);
    for (int c = 0; c < nc; ++c)
      for (int a = 0; a < 3; ++a) vcell[fill[m->cells[3 * (size_t)c + a]]++] = c;
  }
  std::vector<int32_t> sptr(nv + 1, 0), scol;
  {
    std::vector<std::vector<int32_t>> rows(nv);
    mx_par_for(nv, [&](int64_t a, int64_t b) {
      for (int64_t v = a; v < b; ++v) {
        auto& r = rows[v];
        for (int64_t q = vptr[v]; q < vptr[v + 1]; ++q)
          for (int k = 0; k < 3; ++k) r.push_back(m->cells[3 * (size_t)vcell[q] + k]);
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
      }
    });
    for (int v = 0; v < nv; ++v) sptr[v + 1] = sptr[v] + (int32_t)rows[v].size();
    scol.resize(sptr[nv]);
    for (int v = 0; v < nv; ++v) std::copy(rows[v].begin(), rows[v].end(), scol.begin() + sptr[v]);
  }
  const int64_t nnz_s = sptr[nv];
  if (9 * nnz_s > 0x7fffffff) {
    h->err = "mixed matrix exceeds int32 nnz";
    return PGX_EINVAL;
  }
  const int64_t tot = 9 * nnz_s;
  h->nnz = tot;
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(ntot + 1, 0);
  col.resize(tot);
  std::vector<uint8_t> kind(tot);
  for (int fr = 0; fr < 3; ++fr)
    for (int v = 0; v < nv; ++v) {
      const int64_t r = (int64_t)fr * nv + v;
      const int len = sptr[v + 1] - sptr[v];
      rowptr[r + 1] = 3 * len;
    }
  for (int64_t r = 0; r < ntot; ++r) rowptr[r + 1] += rowptr[r];
  for (int fr = 0; fr < 3; ++fr)
    for (int v = 0; v < nv; ++v) {
      const int64_t r = (int64_t)fr * nv + v;
      const int len = sptr[v + 1] - sptr[v];
      for (int fc = 0; fc < 3; ++fc)
        for (int k = 0; k < len; ++k) {
          const int32_t j = scol[sptr[v] + k];
          const int64_t e = rowptr[r] + (int64_t)fc * len + k;
          col[e] = fc * nv + j;
          uint8_t t;
          if (fr == 0 && fc == 0)
            t = (hmask[v] || hmask[j]) ? ((v == j && hmask[v]) ? 3 : 4) : 0;
          else if (fr == 0)
            t = (fc == 2 && !hmask[v]) ? 1 : 4;  // (u,T) is structurally zero, (u,psi) = M
          else if (fr == 1)
            t = fc == 0 ? 4 : fc == 1 ? 1 : 2;   // (T,u) zero, (T,T) constant, (T,psi) per step
          else
            t = fc == 0 ? (hmask[j] ? 4 : 1) : fc == 1 ? 1 : 2;
          kind[e] = t;
        }
    }
  auto find = [&](int fr, int32_t v, int fc, int32_t j) -> int32_t {
    const int len = sptr[v + 1] - sptr[v];
    const int32_t* b = scol.data() + sptr[v];
    const int k = (int)(std::lower_bound(b, b + len, j) - b);
    return (int32_t)(rowptr[(int64_t)fr * nv + v] + (int64_t)fc * len + k);
  };
  // destination tables, slot-major like the stashes: table[slot * nc + cell]
  std::vector<int32_t> d45((size_t)nc * 45), d18((size_t)nc * 18), d9((size_t)nc * 9);
  mx_par_for(nc, [&](int64_t a0, int64_t b0) {
    const int blk_r[5] = {0, 0, 1, 2, 2}, blk_c[5] = {0, 2, 1, 0, 1};
    for (int64_t c = a0; c < b0; ++c) {
      const int32_t* cv = m->cells + 3 * (size_t)c;
      for (int k = 0; k < 5; ++k)
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) d45[(size_t)(k * 9 + a * 3 + b) * nc + (size_t)c] = find(blk_r[k], cv[a], blk_c[k], cv[b]);
      for (int a = 0; a < 3; ++a) {
        for (int fld = 0; fld < 3; ++fld) d9[(size_t)(fld * 3 + a) * nc + (size_t)c] = fld * nv + cv[a];
        for (int b = 0; b < 3; ++b) {
          d18[(size_t)(a * 3 + b) * nc + (size_t)c] = find(1, cv[a], 2, cv[b]);
          d18[(size_t)(9 + a * 3 + b) * nc + (size_t)c] = find(2, cv[a], 2, cv[b]);
        }
      }
    }
  });
  std::vector<int32_t> nod(ntot);
  for (int v = 0; v < nv; ++v) nod[v] = nod[(size_t)nv + v] = nod[2 * (size_t)nv + v] = v;
  MXHIP(hipStreamCreate(&h->st));
  pgx_nd_matrix A{};
  A.n = ntot;
  A.rowptr = rowptr.data();
  A.col = col.data();
  A.n_nodes = nv;
  A.node_of_dof = nod.data();
  A.dim = 2;
  A.node_coords = m->coords;
  A.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) A.leaf_nodes = atoi(e);
  int rc = pgx_nd_create(&A, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  MXALLOC(h->coords, 2 * (size_t)nv);
  MXALLOC(h->cells, 3 * (size_t)nc);
  MXALLOC(h->mask, nv);
  MXALLOC(h->stash, 18 * (size_t)nc);
  MXALLOC(h->rowptr, ntot + 1);
  MXALLOC(h->col, tot);
  MXALLOC(h->kind, tot);
  MXALLOC(h->Jc, tot);
  MXALLOC(h->Jv, tot);
  if ((rc = mx_alloc_state(h))) return rc;
  MXHIP(hipMemcpy(h->coords, m->coords, sizeof(double) * 2 * nv, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->cells, m->cells, sizeof(int32_t) * 3 * (size_t)nc, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->mask, hmask.data(), nv, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (ntot + 1), hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->col, col.data(), sizeof(int32_t) * tot, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->kind, kind.data(), tot, hipMemcpyHostToDevice));
  MXHIP(hipMemsetAsync(h->Jc, 0, sizeof(double) * tot, h->st));
  {
    std::string e1 = pgx_scatter_build(d9.data(), (int64_t)9 * nc, ntot, h->allocs, &h->sc_res);
    if (e1.empty()) e1 = pgx_scatter_build(d18.data(), (int64_t)18 * nc, tot, h->allocs, &h->sc_psi);
    if (!e1.empty()) {
      h->err = e1;
      return PGX_ENOMEM;
    }
  }
  {  // constant blocks, once, deterministic: table and stash are temporary
    std::vector<void*> tmp;
    PgxScatter sc_c;
    std::string e1 = pgx_scatter_build(d45.data(), (int64_t)45 * nc, tot, tmp, &sc_c);
    double* st45 = nullptr;
    hipError_t e = e1.empty() ? hipMalloc((void**)&st45, sizeof(double) * 45 * (size_t)nc) : hipErrorOutOfMemory;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_qv_const, dim3((nc + 127) / 128), dim3(128), 0, h->st, nc, h->cells, h->coords, h->beta, h->Q, st45);
      pgx_scatter_run(h->st, sc_c, st45, 1.0, 0, h->Jc);
      e = hipStreamSynchronize(h->st);
    }
    if (st45) hipFree(st45);
    for (void* q : tmp) hipFree(q);
    if (e != hipSuccess) {
      h->err = std::string("constant Jacobian blocks: ") + (e1.empty() ? hipGetErrorString(e) : e1.c_str());
      return PGX_EHIP;
    }
  }
  return PGX_OK;
}

extern "C" int pgx_qvi_create(const pgx_mesh* m, const pgx_qvi_problem* p, int device, pgx_qvi_handle** out) {
  if (!m || !p || !out || !m->coords || !m->cells || m->n_vertices <= 0 || m->n_cells <= 0 || !p->qpts || !p->qwts ||
      p->nq <= 0 || p->nq > QV_MAXQ || (p->n_bc > 0 && !p->bc_dofs) || !(p->knee > 0.0)) {
    g_qvi_error = "pgx_qvi_create: bad arguments";
    return PGX_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_qvi_error = "pgx_qvi_create: no usable GPU (there is no CPU fallback)";
    return PGX_ENODEV;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_qvi_error = "hipSetDevice failed";
    return PGX_EHIP;
  }
  pgx_qvi_handle* h = new pgx_qvi_handle();
  h->device = device;
  int rc = qvi_create_impl(h, m, p);
  if (rc) {
    g_qvi_error = h->err;
    pgx_qvi_destroy(h);
    return rc;
  }
  *out = h;
  return PGX_OK;
}

#define QVNEED(h)              \
  if (!(h)) return PGX_EINVAL; \
  if (hipSetDevice((h)->device) != hipSuccess) return PGX_EHIP

extern "C" int pgx_qvi_num_dofs(const pgx_qvi_handle* h, int64_t* ntot) {
  if (!h || !ntot) return PGX_EINVAL;
  *ntot = h->ntot;
  return PGX_OK;
}
extern "C" int pgx_qvi_set_state(pgx_qvi_handle* h, const double* x) {
  QVNEED(h);
  return mx_in(h, h->x, x);
}
extern "C" int pgx_qvi_get_state(pgx_qvi_handle* h, double* x) {
  QVNEED(h);
  return mx_out(h, x, h->x);
}
extern "C" int pgx_qvi_set_prev(pgx_qvi_handle* h, const double* x) {
  QVNEED(h);
  return mx_in(h, h->xk, x);
}
extern "C" int pgx_qvi_get_prev(pgx_qvi_handle* h, double* x) {
  QVNEED(h);
  return mx_out(h, x, h->xk);
}
extern "C" int pgx_qvi_advance_prev(pgx_qvi_handle* h) {
  QVNEED(h);
  MXHIP(hipMemcpyAsync(h->xk, h->x, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_qvi_set_alpha(pgx_qvi_handle* h, double a) {
  QVNEED(h);
  if (!(a > 0.0) || !std::isfinite(a)) {
    h->err = "alpha must be positive and finite";
    return PGX_EINVAL;
  }
  h->alpha = a;
  h->jac_valid = false;
  return PGX_OK;
}
extern "C" int pgx_qvi_residual(pgx_qvi_handle* h, const double* x, double* F, double* fnorm) {
  QVNEED(h);
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
extern "C" int pgx_qvi_jacobian_fill(pgx_qvi_handle* h, const double* x) {
  QVNEED(h);
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
extern "C" int pgx_qvi_csr_export(pgx_qvi_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col,
                                  double* vals) {
  QVNEED(h);
  if (nrows) *nrows = h->ntot;
  if (nnz) *nnz = h->nnz;
  if (rowptr) std::copy(h->h_rowptr.begin(), h->h_rowptr.end(), rowptr);
  if (col) std::copy(h->h_col.begin(), h->h_col.end(), col);
  if (vals) {
    if (!h->jac_valid) {
      h->err = "pgx_qvi_csr_export: no Jacobian has been filled";
      return PGX_ESTATE;
    }
    MXHIP(hipMemcpy(vals, h->Jv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  }
  return PGX_OK;
}
extern "C" int pgx_qvi_spmv(pgx_qvi_handle* h, const double* x, double* y) {
  QVNEED(h);
  if (!x || !y) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_qvi_spmv: no Jacobian has been filled";
    return PGX_ESTATE;
  }
  int rc = mx_in(h, h->r, x);
  if (rc) return rc;
  mx_spmv_dev(h, h->r, h->z);
  return mx_out(h, y, h->z);
}
extern "C" int pgx_qvi_newton_solve(pgx_qvi_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its) {
  QVNEED(h);
  if (!opts) return PGX_EINVAL;
  return opts->linesearch == 1 ? mx_newton_solve_bt(h, opts, reason, its, lin_its) : mx_newton_solve(h, opts, reason, its, lin_its);
}
extern "C" int pgx_qvi_h1_increment(pgx_qvi_handle* h, double* out) {
  QVNEED(h);
  if (!out) return PGX_EINVAL;
  hipLaunchKernelGGL(k_qv_h1, dim3(MX_RED), dim3(256), 0, h->st, h->nc, h->cells, h->coords, h->x, h->xk, h->Q, h->partials);
  hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
  MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  *out = std::sqrt(std::max(h->h_out[0], 0.0));
  return PGX_OK;
}
extern "C" int pgx_qvi_profile(pgx_qvi_handle* h, int enable, double ms[6]) {
  QVNEED(h);
  pgx_nd_timing(h->lu, enable, nullptr, nullptr);
  return mx_profile(h, enable, ms);
}
