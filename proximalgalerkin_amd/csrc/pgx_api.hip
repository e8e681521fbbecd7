// Host side of libpgx.so: plan building (CSR pattern, inverted vertex->cell lists), multigrid
// hierarchy, FGMRES, SNES-mirroring Newton driver, and the extern "C" ABI of include/pgx.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pgx.h"
#include "pgx_internal.h"

static std::string g_create_error;

struct pgx_handle {
  int device = 0;
  hipStream_t st = nullptr;
  std::string err;
  int n = 0, nc = 0, nx = 0, ny = 0, nnz = 0;
  bool structured = false;
  QuadTab q{};
  double f = 0.0, alpha = 1.0;
  // mesh + problem data
  double* coords = nullptr;
  int32_t* cells = nullptr;
  uint8_t* mask = nullptr;
  double *gbc = nullptr, *bphi = nullptr;
  // CSR pattern + inverted lists
  int32_t *rowptr = nullptr, *colm = nullptr, *v2c_ptr = nullptr, *v2c_ent = nullptr, *v2c_pos = nullptr;
  std::vector<int32_t> h_rowptr, h_col;
  size_t fill_lds = 0;
  double *Kv = nullptr, *Mv = nullptr, *Dv = nullptr;
  bool jac_valid = false;
  // state
  double *x = nullptr, *xk = nullptr, *F = nullptr, *dx = nullptr, *xw = nullptr, *rhs = nullptr;
  // Krylov workspace
  int restart = 0;
  double *V = nullptr, *Z = nullptr, *w = nullptr, *d_small = nullptr, *partials = nullptr;
  double* h_small = nullptr;  // pinned
  // multigrid
  std::vector<GridLevel> lev;
  double *tmp_u = nullptr, *tmp_p = nullptr, *res_u = nullptr, *res_p = nullptr;  // level-0 scratch (each n)
  int coarse_sweeps = 4;  // prototype (oracle/krylov_proto.py): 2..60 sweeps give identical Krylov counts
  int tail_start = -1;  // first level handled by the fused k_mg_tail launch (-1: none)
  int xcd_remap = 2;    // bit 1 of the `first` kernel argument; PGX_XCD_REMAP=0 disables (A/B: +1..3 %)
  int tail_verts = 1100;
  int spmv_stream = 1;  // PGX_SPMV_STREAM=0: 8-lanes-per-row kernel instead of the CSR-stream kernel
  int fused_legs = 1;   // PGX_FUSED_LEGS=0: one launch per sweep / residual / restriction / prolongation
  int fused_min = 500000;  // fused legs only pay on levels large enough to hide their 3-phase latency
  TailArgs tail{};
  // observables
  double *obs_partials = nullptr, *d_out6 = nullptr;
  int obs_blocks = 0;
  // profiling
  bool prof = false;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<void*> allocs;
};

#define HIPCHK(call)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
      return PGX_EHIP;                                                                          \
    }                                                                                           \
  } while (0)

template <typename T>
static int dalloc(pgx_handle* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return PGX_ENOMEM;
  }
  h->allocs.push_back(q);
  *p = (T*)q;
  return PGX_OK;
}
#define DALLOC(p, count)                         \
  do {                                           \
    int rc_ = dalloc(h, &(p), (size_t)(count));  \
    if (rc_) return rc_;                         \
  } while (0)

struct PhaseTimer {
  pgx_handle* h;
  int slot;
  PhaseTimer(pgx_handle* h_, int s) : h(h_), slot(s) {
    if (h->prof) hipEventRecord(h->e0, h->st);
  }
  ~PhaseTimer() {
    if (h->prof) {
      hipEventRecord(h->e1, h->st);
      hipEventSynchronize(h->e1);
      float ms = 0;
      hipEventElapsedTime(&ms, h->e0, h->e1);
      h->ms[slot] += ms;
    }
  }
};

extern "C" void pgx_default_opts(pgx_snes_opts* o) {
  o->snes_rtol = 1e-8;
  o->snes_atol = 1e-50;
  o->snes_stol = 1e-8;
  o->snes_divtol = 1e4;
  o->snes_max_it = 50;
  o->ksp_rtol = 1e-9;  // final u moves 9e-14 (bar 1e-10) vs a 1e-13 solve at 2048^2: DESIGN.md section 3
  o->ksp_max_it = 200;
  o->ksp_restart = 30;
  o->mg_nu = 2;
  o->mg_omega = 0.8;
  o->monitor = 0;
}

extern "C" const char* pgx_last_error(const pgx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// ------------------------------------------------------------------------------------------------
// plan building on the host (setup only; not on the hot path)
// ------------------------------------------------------------------------------------------------
static int build_plan(pgx_handle* h, const pgx_mesh* m, const std::vector<uint8_t>& hmask) {
  const int n = m->n_vertices, nc = m->n_cells;
  std::vector<int32_t> vptr(n + 1, 0);
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 3; ++a) {
      const int v = m->cells[3 * c + a];
      if (v < 0 || v >= n) {
        h->err = "cell vertex id out of range";
        return PGX_EINVAL;
      }
      vptr[v + 1]++;
    }
  for (int i = 0; i < n; ++i) vptr[i + 1] += vptr[i];
  std::vector<int32_t> vent(vptr[n]), fillp(vptr.begin(), vptr.end() - 1);
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 3; ++a) vent[fillp[m->cells[3 * c + a]]++] = c * 4 + a;
  // rows: sorted unique neighbour vertices
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(n + 1, 0);
  col.clear();
  col.reserve((size_t)n * 7);
  std::vector<int32_t> tmp;
  for (int i = 0; i < n; ++i) {
    tmp.clear();
    for (int k = vptr[i]; k < vptr[i + 1]; ++k) {
      const int c = vent[k] >> 2;
      tmp.push_back(m->cells[3 * c]);
      tmp.push_back(m->cells[3 * c + 1]);
      tmp.push_back(m->cells[3 * c + 2]);
    }
    if (tmp.empty()) tmp.push_back(i);  // isolated vertex: keep a diagonal
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if (tmp.size() > 255) {
      h->err = "vertex degree > 255 unsupported";
      return PGX_EINVAL;
    }
    col.insert(col.end(), tmp.begin(), tmp.end());
    rowptr[i + 1] = (int32_t)col.size();
  }
  if (col.size() > 0x7fffffffu) {
    h->err = "nnz overflows int32";
    return PGX_EINVAL;
  }
  h->nnz = (int)col.size();
  std::vector<int32_t> vpos(vent.size());
  for (int i = 0; i < n; ++i) {
    const int32_t* rb = col.data() + rowptr[i];
    const int32_t* re = col.data() + rowptr[i + 1];
    for (int k = vptr[i]; k < vptr[i + 1]; ++k) {
      const int c = vent[k] >> 2;
      int pos = 0;
      for (int b = 0; b < 3; ++b) {
        const int p = (int)(std::lower_bound(rb, re, m->cells[3 * c + b]) - rb);
        pos |= p << (8 * b);
      }
      vpos[k] = pos;
    }
  }
  size_t maxlen = 0;
  for (int i0 = 0; i0 < n; i0 += PGX_BLOCK) {
    const int i1 = std::min(i0 + PGX_BLOCK, n);
    maxlen = std::max(maxlen, (size_t)(rowptr[i1] - rowptr[i0]));
  }
  h->fill_lds = maxlen * sizeof(double);
  if (h->fill_lds > 150 * 1024) {
    h->err = "row block too dense for the LDS-staged fill";
    return PGX_EINVAL;
  }
  if (h->structured) {
    const int sx = h->nx + 1;
    for (int i = 0; i < n; ++i)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int o = col[k] - i;
        if (!(o == 0 || o == 1 || o == -1 || o == sx || o == -sx || o == sx + 1 || o == -sx - 1)) {
          h->err = "mesh flagged structured is not the right-diagonal triangulation (vertex v=j*(nx+1)+i)";
          return PGX_EINVAL;
        }
      }
  }
  std::vector<int32_t> colm(col.size());
  for (size_t k = 0; k < col.size(); ++k) colm[k] = col[k] | (hmask[col[k]] ? (int32_t)0x80000000 : 0);
  DALLOC(h->rowptr, n + 1);
  DALLOC(h->colm, colm.size());
  DALLOC(h->v2c_ptr, n + 1);
  DALLOC(h->v2c_ent, vent.size());
  DALLOC(h->v2c_pos, vpos.size());
  HIPCHK(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->colm, colm.data(), sizeof(int32_t) * colm.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->v2c_ptr, vptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->v2c_ent, vent.data(), sizeof(int32_t) * vent.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->v2c_pos, vpos.data(), sizeof(int32_t) * vpos.size(), hipMemcpyHostToDevice));
  return PGX_OK;
}

// Is the stencil the same at every interior vertex (uniform grid)?  Then interior rows take their K and M
// coefficients from kernel arguments.  Setup-time host check on a downloaded copy.
static int detect_uniform(pgx_handle* h, GridLevel& L) {
  L.uniform = 0;
  if (L.nx < 2 || L.ny < 2) return PGX_OK;
  std::vector<double> K((size_t)7 * L.n), M((size_t)7 * L.n);
  HIPCHK(hipMemcpyAsync(K.data(), L.K, K.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipMemcpyAsync(M.data(), L.M, M.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  const int sx = L.nx + 1;
  const int v0 = sx + 1;
  double kmax = 0, mmax = 0;
  for (int s = 0; s < 7; ++s) {
    L.Kc[s] = K[(size_t)s * L.n + v0];
    L.Mc[s] = M[(size_t)s * L.n + v0];
    kmax = std::max(kmax, std::fabs(L.Kc[s]));
    mmax = std::max(mmax, std::fabs(L.Mc[s]));
  }
  bool same = true;
  for (int j = 1; j < L.ny && same; ++j)
    for (int i = 1; i < L.nx && same; ++i) {
      const size_t v = (size_t)j * sx + i;
      for (int s = 0; s < 7; ++s)
        if (std::fabs(K[(size_t)s * L.n + v] - L.Kc[s]) > 1e-11 * kmax ||
            std::fabs(M[(size_t)s * L.n + v] - L.Mc[s]) > 1e-11 * mmax) {
          same = false;
          break;
        }
    }
  L.uniform = same ? 1 : 0;
  return PGX_OK;
}

static int build_multigrid(pgx_handle* h) {
  GridLevel L0{};
  L0.nx = h->nx;
  L0.ny = h->ny;
  L0.n = h->n;
  L0.mask = h->mask;
  h->lev.push_back(L0);
  if (!h->structured) return PGX_OK;
  DALLOC(h->lev[0].K, (size_t)7 * h->n);
  DALLOC(h->lev[0].M, (size_t)7 * h->n);
  DALLOC(h->lev[0].Dh, (size_t)4 * h->n);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Kv, h->lev[0].K);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Mv, h->lev[0].M);
  int rc = detect_uniform(h, h->lev[0]);
  if (rc) return rc;
  int nx = h->nx, ny = h->ny;
  while (nx % 2 == 0 && ny % 2 == 0 && nx > 2 && ny > 2) {
    nx /= 2;
    ny /= 2;
    GridLevel L{};
    L.nx = nx;
    L.ny = ny;
    L.n = (nx + 1) * (ny + 1);
    DALLOC(L.K, (size_t)7 * L.n);
    DALLOC(L.M, (size_t)7 * L.n);
    DALLOC(L.Dh, (size_t)4 * L.n);
    DALLOC(L.mask, L.n);
    DALLOC(L.xu, L.n);
    DALLOC(L.xp, L.n);
    DALLOC(L.xu2, L.n);
    DALLOC(L.xp2, L.n);
    DALLOC(L.bu, L.n);
    DALLOC(L.bp, L.n);
    DALLOC(L.ru, L.n);
    DALLOC(L.rp, L.n);
    const GridLevel& Fl = h->lev.back();
    pgxk_coarse_mask(h->st, L, L.mask, Fl);
    pgxk_rap7(h->st, Fl, Fl.K, L, L.K);
    pgxk_rap7(h->st, Fl, Fl.M, L, L.M);
    rc = detect_uniform(h, L);
    if (rc) return rc;
    h->lev.push_back(L);
  }
  // fused tail: every level with at most PGX_TAIL_VERTS vertices (and at most PGX_TAIL_MAX of them)
  const int nl = (int)h->lev.size();
  for (int l = 1; l < nl; ++l)
    if (h->lev[l].n <= h->tail_verts && nl - l <= PGX_TAIL_MAX) {
      h->tail_start = l;
      break;
    }
  if (h->tail_start > 0) {
    h->tail.nlev = nl - h->tail_start;
    for (int l = h->tail_start; l < nl; ++l) {
      const GridLevel& L = h->lev[l];
      TailLevel& T = h->tail.L[l - h->tail_start];
      T.nx = L.nx;
      T.ny = L.ny;
      T.n = L.n;
      T.K = L.K;
      T.M = L.M;
      T.Dh = L.Dh;
      for (int s = 0; s < 7; ++s) {
        T.sc.K[s] = L.Kc[s];
        T.sc.M[s] = L.Mc[s];
      }
      T.sc.uniform = L.uniform;
      T.mask = L.mask;
      T.xu = L.xu;
      T.xp = L.xp;
      T.xu2 = L.xu2;
      T.xp2 = L.xp2;
      T.bu = L.bu;
      T.bp = L.bp;
      T.ru = L.ru;
      T.rp = L.rp;
    }
  }
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int pgx_create(const pgx_mesh* m, const pgx_problem* p, int device, pgx_handle** out) {
  if (!m || !p || !out) {
    g_create_error = "null argument";
    return PGX_EINVAL;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible: libpgx has no CPU fallback";
    return PGX_ENODEV;
  }
  if (device < 0 || device >= ndev) {
    g_create_error = "device index out of range";
    return PGX_EINVAL;
  }
  pgx_handle* h = new pgx_handle();
  if (const char* e = getenv("PGX_XCD_REMAP")) h->xcd_remap = atoi(e) ? 2 : 0;
  if (const char* e = getenv("PGX_TAIL_VERTS")) h->tail_verts = atoi(e);
  if (const char* e = getenv("PGX_FUSED_LEGS")) h->fused_legs = atoi(e);
  if (const char* e = getenv("PGX_SPMV_STREAM")) h->spmv_stream = atoi(e);
  if (const char* e = getenv("PGX_FUSED_MIN")) h->fused_min = atoi(e);
  auto fail = [&](int rc) {
    g_create_error = h->err;
    pgx_destroy(h);
    return rc;
  };
  h->device = device;
  if (p->degree != 1) {
    h->err = "only degree 1 is implemented";
    return fail(PGX_EINVAL);
  }
  if (p->nq < 1 || p->nq > PGX_MAX_NQ || !p->qpts || !p->qwts || !p->phi_q || m->n_vertices < 3 || m->n_cells < 1 ||
      !m->coords || !m->cells || (p->n_bc > 0 && !p->bc_dofs)) {
    h->err = "invalid mesh/problem description";
    return fail(PGX_EINVAL);
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->st) != hipSuccess) {
    h->err = "hipSetDevice/hipStreamCreate failed";
    return fail(PGX_EHIP);
  }
  hipEventCreate(&h->e0);
  hipEventCreate(&h->e1);
  const int n = h->n = m->n_vertices, nc = h->nc = m->n_cells;
  h->f = p->f;
  if (m->structured_nx > 0 && m->structured_ny > 0) {
    if ((int64_t)(m->structured_nx + 1) * (m->structured_ny + 1) != n ||
        (int64_t)2 * m->structured_nx * m->structured_ny != nc) {
      h->err = "structured_nx/ny inconsistent with n_vertices/n_cells";
      return fail(PGX_EINVAL);
    }
    h->structured = true;
    h->nx = m->structured_nx;
    h->ny = m->structured_ny;
  }
  // quadrature tables
  h->q.nq = p->nq;
  memset(h->q.Mref, 0, sizeof(h->q.Mref));
  memset(h->q.mref, 0, sizeof(h->q.mref));
  for (int k = 0; k < p->nq; ++k) {
    const double X = p->qpts[2 * k], Y = p->qpts[2 * k + 1];
    h->q.N[k][0] = 1.0 - X - Y;
    h->q.N[k][1] = X;
    h->q.N[k][2] = Y;
    h->q.w[k] = p->qwts[k];
    for (int a = 0; a < 3; ++a) {
      h->q.mref[a] += h->q.w[k] * h->q.N[k][a];
      for (int b = 0; b < 3; ++b) h->q.Mref[a][b] += h->q.w[k] * h->q.N[k][a] * h->q.N[k][b];
    }
  }
  // Dirichlet data
  std::vector<uint8_t> hmask(n, 0);
  std::vector<double> hg(n, 0.0);
  for (int k = 0; k < p->n_bc; ++k) {
    const int d = p->bc_dofs[k];
    if (d < 0 || d >= n) {
      h->err = "bc dof out of range";
      return fail(PGX_EINVAL);
    }
    hmask[d] = 1;
    hg[d] = p->bc_vals ? p->bc_vals[k] : 0.0;
  }
  int rc;
  auto up = [&]() -> int {
    DALLOC(h->coords, (size_t)2 * n);
    DALLOC(h->cells, (size_t)3 * nc);
    DALLOC(h->mask, n);
    DALLOC(h->gbc, n);
    DALLOC(h->bphi, n);
    HIPCHK(hipMemcpy(h->coords, m->coords, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cells, m->cells, sizeof(int32_t) * 3 * (size_t)nc, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->mask, hmask.data(), n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->gbc, hg.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    // b_phi from phi at quadrature points, then phi_q is dropped (it never changes: obstacle_pg.py:107-111)
    double* phi_q = nullptr;
    const size_t nphi = (size_t)nc * p->nq;
    if (hipMalloc((void**)&phi_q, nphi * sizeof(double)) != hipSuccess) {
      h->err = "hipMalloc(phi_q)";
      return PGX_ENOMEM;
    }
    hipError_t e = hipMemcpy(phi_q, p->phi_q, nphi * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      pgxk_bphi(h->st, nc, n, h->cells, h->coords, phi_q, h->q, h->bphi);
      e = hipStreamSynchronize(h->st);
    }
    hipFree(phi_q);
    if (e != hipSuccess) {
      h->err = std::string("b_phi assembly: ") + hipGetErrorString(e);
      return PGX_EHIP;
    }
    int r = build_plan(h, m, hmask);
    if (r) return r;
    DALLOC(h->Kv, h->nnz);
    DALLOC(h->Mv, h->nnz);
    DALLOC(h->Dv, h->nnz);
    pgxk_fill_rows(h->st, 0, n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                   nullptr, h->q, h->Kv);
    pgxk_fill_rows(h->st, 1, n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                   nullptr, h->q, h->Mv);
    const size_t n2 = 2 * (size_t)n;
    DALLOC(h->x, n2);
    DALLOC(h->xk, n2);
    DALLOC(h->F, n2);
    DALLOC(h->dx, n2);
    DALLOC(h->xw, n2);
    DALLOC(h->rhs, n2);
    HIPCHK(hipMemsetAsync(h->x, 0, n2 * sizeof(double), h->st));
    HIPCHK(hipMemsetAsync(h->xk, 0, n2 * sizeof(double), h->st));
    h->restart = 50;
    DALLOC(h->V, (size_t)(h->restart + 1) * n2);
    DALLOC(h->Z, (size_t)h->restart * n2);
    DALLOC(h->w, n2);
    DALLOC(h->d_small, 4 * (h->restart + 2));
    DALLOC(h->partials, (size_t)PGX_RED_BLOCKS * (h->restart + 2));
    HIPCHK(hipHostMalloc((void**)&h->h_small, sizeof(double) * 4 * (h->restart + 2)));
    DALLOC(h->tmp_u, n);
    DALLOC(h->tmp_p, n);
    DALLOC(h->res_u, n);
    DALLOC(h->res_p, n);
    h->obs_blocks = pgxk_observables_blocks(nc);
    DALLOC(h->obs_partials, (size_t)h->obs_blocks * 6);
    DALLOC(h->d_out6, 6);
    r = build_multigrid(h);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(h->st));
    return PGX_OK;
  };
  rc = up();
  if (rc) return fail(rc);
  *out = h;
  return PGX_OK;
}

extern "C" void pgx_destroy(pgx_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->st) hipStreamSynchronize(h->st);
  for (void* p : h->allocs) hipFree(p);
  if (h->h_small) hipHostFree(h->h_small);
  if (h->e0) hipEventDestroy(h->e0);
  if (h->e1) hipEventDestroy(h->e1);
  if (h->st) hipStreamDestroy(h->st);
  delete h;
}

// ------------------------------------------------------------------------------------------------
// state access
// ------------------------------------------------------------------------------------------------
#define NEED(hh)          \
  if (!(hh)) return PGX_EINVAL; \
  hipSetDevice((hh)->device)

extern "C" int pgx_num_dofs(const pgx_handle* h, int64_t* nd) {
  if (!h || !nd) return PGX_EINVAL;
  *nd = 2 * (int64_t)h->n;
  return PGX_OK;
}
static int copy_in(pgx_handle* h, double* dst, const double* src) {
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * 2 * (size_t)h->n, hipMemcpyHostToDevice, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
static int copy_out(pgx_handle* h, double* dst, const double* src) {
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * 2 * (size_t)h->n, hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_set_state(pgx_handle* h, const double* x) {
  NEED(h);
  if (!x) return PGX_EINVAL;
  return copy_in(h, h->x, x);
}
extern "C" int pgx_get_state(pgx_handle* h, double* x) {
  NEED(h);
  if (!x) return PGX_EINVAL;
  return copy_out(h, x, h->x);
}
extern "C" int pgx_set_prev(pgx_handle* h, const double* x) {
  NEED(h);
  if (!x) return PGX_EINVAL;
  return copy_in(h, h->xk, x);
}
extern "C" int pgx_get_prev(pgx_handle* h, double* x) {
  NEED(h);
  if (!x) return PGX_EINVAL;
  return copy_out(h, x, h->xk);
}
extern "C" int pgx_advance_prev(pgx_handle* h) {
  NEED(h);
  HIPCHK(hipMemcpyAsync(h->xk, h->x, sizeof(double) * 2 * (size_t)h->n, hipMemcpyDeviceToDevice, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_zero_state(pgx_handle* h) {
  NEED(h);
  HIPCHK(hipMemsetAsync(h->x, 0, sizeof(double) * 2 * (size_t)h->n, h->st));
  HIPCHK(hipMemsetAsync(h->xk, 0, sizeof(double) * 2 * (size_t)h->n, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_set_alpha(pgx_handle* h, double a) {
  if (!h || !(a > 0.0)) return PGX_EINVAL;
  h->alpha = a;
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------------
static int dev_norm(pgx_handle* h, const double* v, double* out) {
  pgxk_multidot(h->st, 2 * (size_t)h->n, 1, v, 0, v, h->partials, h->d_small);
  HIPCHK(hipMemcpyAsync(h->h_small, h->d_small, sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  *out = std::sqrt(h->h_small[0]);
  return PGX_OK;
}

static void residual_dev(pgx_handle* h, const double* x, double* F) {
  PhaseTimer t(h, 0);
  pgxk_residual(h->st, h->nc, h->n, h->cells, h->coords, h->mask, h->gbc, h->bphi, x, h->xk, h->alpha, h->f, h->q, F);
}

static void jacobian_dev(pgx_handle* h, const double* x) {
  {
    PhaseTimer t(h, 1);
    pgxk_fill_rows(h->st, 2, h->n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                   x + h->n, h->q, h->Dv);
  }
  if (h->structured) {
    PhaseTimer t(h, 2);
    pgxk_csr_to_stencil_h(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Dv, h->lev[0].Dh);
    for (size_t l = 1; l < h->lev.size(); ++l)
      pgxk_rap7h(h->st, h->lev[l - 1], h->lev[l - 1].Dh, h->lev[l], h->lev[l].Dh);
  }
  h->jac_valid = true;
}

// y = J x on device vectors of length 2n
static void spmv_dev(pgx_handle* h, const double* x, double* y) {
  if (h->spmv_stream && 2 * h->fill_lds <= 64 * 1024)
    pgxk_bspmv_stream(h->st, h->n, h->fill_lds, h->rowptr, h->colm, h->Kv, h->Mv, h->Dv, h->alpha, h->mask, x,
                      x + h->n, h->xcd_remap ? 1 : 0, y, y + h->n);
  else
    pgxk_bspmv(h->st, 0, h->n, h->rowptr, h->colm, h->Kv, h->Mv, h->Dv, h->alpha, x, x + h->n, nullptr, nullptr, 0.0,
               h->xcd_remap, y, y + h->n);
}

static void level_apply(pgx_handle* h, int l, int mode, const double* xu, const double* xp, const double* bu,
                        const double* bp, double omega, int first, double* yu, double* yp) {
  if (l == 0 && !h->structured)  // general mesh: the block-CSR kernel is also the (single-level) smoother
    pgxk_bspmv(h->st, mode, h->n, h->rowptr, h->colm, h->Kv, h->Mv, h->Dv, h->alpha, xu, xp, bu, bp, omega,
               first | h->xcd_remap, yu, yp);
  else
    pgxk_st_apply(h->st, mode, h->lev[l], h->alpha, xu, xp, bu, bp, omega, first | h->xcd_remap, yu, yp);
}

// one V(nu,nu) cycle for J_l x = b, zero initial guess, result in (outu,outp)
static void vcycle(pgx_handle* h, int l, const double* bu, const double* bp, double* outu, double* outp, int nu,
                   double omega) {
  GridLevel& L = h->lev[l];
  if (l > 0 && l == h->tail_start) {  // all remaining levels in ONE launch (k_mg_tail); result in L.xu/L.xp
    h->tail.nu = nu;
    h->tail.omega = omega;
    h->tail.alpha = h->alpha;
    h->tail.coarse_sweeps = h->coarse_sweeps;
    pgxk_mg_tail(h->st, h->tail);
    return;
  }
  const bool last = (l + 1 == (int)h->lev.size());
  double* Au = outu;
  double* Ap = outp;
  double* Bu = (l == 0) ? h->tmp_u : L.xu2;
  double* Bp = (l == 0) ? h->tmp_p : L.xp2;
  double* ru = (l == 0) ? h->res_u : L.ru;
  double* rp = (l == 0) ? h->res_p : L.rp;
  if (h->structured && !last && nu == 2 && h->fused_legs && L.n >= h->fused_min) {
    // 3 launches per level: S(S(0)) | P^T(b - Jx) | S(S(x + P x_c))   (pgx_kernels.hip, "Fused V-cycle legs")
    GridLevel& C = h->lev[l + 1];
    const int remap = h->xcd_remap ? 1 : 0;
    pgxk_st_smooth2(h->st, 0, L, h->alpha, nullptr, nullptr, nullptr, nullptr, nullptr, bu, bp, omega, remap, Bu, Bp);
    pgxk_st_resid_restrict(h->st, L, h->alpha, Bu, Bp, bu, bp, C, remap, C.bu, C.bp);
    vcycle(h, l + 1, C.bu, C.bp, C.xu, C.xp, nu, omega);
    pgxk_st_smooth2(h->st, 1, L, h->alpha, Bu, Bp, &C, C.xu, C.xp, bu, bp, omega, remap, Au, Ap);
    return;
  }
  const int total = last ? (h->lev.size() == 1 ? 2 * nu : h->coarse_sweeps) : 2 * nu;
  bool toA = (total % 2) == 1;  // alternate targets so that the final sweep lands in A
  const double *cu = nullptr, *cp = nullptr;
  auto sweep = [&](int first) {
    double* tu = toA ? Au : Bu;
    double* tp = toA ? Ap : Bp;
    level_apply(h, l, 2, cu, cp, bu, bp, omega, first, tu, tp);
    cu = tu;
    cp = tp;
    toA = !toA;
  };
  if (last) {
    for (int s = 0; s < total; ++s) sweep(s == 0);
    return;
  }
  for (int s = 0; s < nu; ++s) sweep(s == 0);
  level_apply(h, l, 1, cu, cp, bu, bp, 0.0, 0, ru, rp);
  GridLevel& C = h->lev[l + 1];
  pgxk_restrict(h->st, L, ru, rp, C, C.bu, C.bp);
  vcycle(h, l + 1, C.bu, C.bp, C.xu, C.xp, nu, omega);
  pgxk_prolong_add(h->st, C, C.xu, C.xp, L, (double*)cu, (double*)cp);
  for (int s = 0; s < nu; ++s) sweep(0);
}

// FGMRES(restart) on J dx = b, right-preconditioned by one V-cycle; CGS2 orthogonalisation with
// batched device dot products; Givens rotations on the host (one small D2H copy + sync per iteration).
static int fgmres(pgx_handle* h, const double* b, double* x, const pgx_snes_opts* o, int* its_out, double* relres) {
  const size_t n2 = 2 * (size_t)h->n;
  const int m = std::min(std::max(o->ksp_restart, 1), h->restart);
  const int n = h->n;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m);
  double bnorm;
  int rc = dev_norm(h, b, &bnorm);
  if (rc) return rc;
  pgxk_set(h->st, n2, 0.0, x);
  *its_out = 0;
  *relres = 0.0;
  if (bnorm == 0.0) return PGX_OK;
  if (!std::isfinite(bnorm)) {
    *relres = bnorm;
    return PGX_OK;
  }
  const double target = o->ksp_rtol * bnorm;
  int its = 0;
  double res = bnorm;
  double prev_cycle_res = bnorm;
  bool first_cycle = true;
  while (true) {
    double beta;
    if (first_cycle && its >= o->ksp_max_it) break;
    if (first_cycle) {
      beta = bnorm;
      pgxk_scale_copy(h->st, n2, 1.0 / beta, b, h->V);
    } else {
      // r = b - J x
      PhaseTimer t(h, 3);
      spmv_dev(h, x, h->w);
      pgxk_scale_copy(h->st, n2, -1.0, h->w, h->w);
      pgxk_axpy(h->st, n2, 1.0, b, h->w);
      rc = dev_norm(h, h->w, &beta);
      if (rc) return rc;
      res = beta;
      if (o->monitor > 1) printf("      ksp true residual after cycle: %.6e (rel %.3e)\n", beta, beta / bnorm);
      if (beta <= target) break;
      // attainable-accuracy exit: a full restart cycle that gains < 10x once we are at LU-level residuals
      if (beta > 0.1 * prev_cycle_res && beta <= 1e-7 * bnorm) break;
      if (its >= o->ksp_max_it) break;
      prev_cycle_res = beta;
      pgxk_scale_copy(h->st, n2, 1.0 / beta, h->w, h->V);
    }
    first_cycle = false;
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    int j = 0;
    for (; j < m && its < o->ksp_max_it; ++j) {
      double* vj = h->V + (size_t)j * n2;
      double* zj = h->Z + (size_t)j * n2;
      {
        PhaseTimer t(h, 4);
        vcycle(h, 0, vj, vj + n, zj, zj + n, o->mg_nu, o->mg_omega);
      }
      {
        PhaseTimer t(h, 3);
        spmv_dev(h, zj, h->V + (size_t)(j + 1) * n2);
      }
      // w = J z_j was written straight into the V_{j+1} slot.  CGS2 (classical Gram-Schmidt, always two
      // passes: one pass loses orthogonality on these ill-conditioned systems and the true residual stalls).
      // Each pass is ONE batched dot kernel [h; ww] = [V_0..V_j, w]^T w plus one batched axpy; the norm of the
      // result comes from Pythagoras on the second pass (|w''|^2 = ww' - |h2|^2, cancellation-free because
      // the second pass removes almost nothing), so no separate norm kernel.
      double* wj = h->V + (size_t)(j + 1) * n2;
      double hn = 0.0;
      {
        PhaseTimer t(h, 5);
        double* d_h1 = h->d_small;
        double* d_h2 = h->d_small + (m + 2);
        pgxk_multidot(h->st, n2, j + 1, h->V, n2, wj, h->partials, d_h1);
        pgxk_multiaxpy(h->st, n2, j + 1, h->V, n2, d_h1, wj);
        pgxk_multidot(h->st, n2, j + 2, h->V, n2, wj, h->partials, d_h2);
        pgxk_multiaxpy(h->st, n2, j + 1, h->V, n2, d_h2, wj);
        HIPCHK(hipMemcpyAsync(h->h_small, h->d_small, sizeof(double) * (2 * (m + 2)), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        double hh = 0.0;
        for (int i = 0; i <= j; ++i) {
          const double h2 = h->h_small[(m + 2) + i];
          H[(size_t)i * m + j] = h->h_small[i] + h2;
          hh += h2 * h2;
        }
        hn = std::sqrt(std::max(h->h_small[(m + 2) + j + 1] - hh, 0.0));
      }
      H[(size_t)(j + 1) * m + j] = hn;
      for (int i = 0; i < j; ++i) {
        const double t1 = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = t1;
      }
      const double a = H[(size_t)j * m + j], bb = H[(size_t)(j + 1) * m + j];
      const double rr = std::hypot(a, bb);
      cs[j] = rr == 0.0 ? 1.0 : a / rr;
      sn[j] = rr == 0.0 ? 0.0 : bb / rr;
      H[(size_t)j * m + j] = rr;
      H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      ++its;
      if (o->monitor > 1) printf("      ksp %3d  rnorm %.6e  rel %.3e\n", its, res, res / bnorm);
      if (!std::isfinite(res)) {
        *its_out = its;
        *relres = res / bnorm;
        return PGX_OK;
      }
      if (res <= target || hn == 0.0) {
        ++j;
        break;
      }
      pgxk_scale_copy(h->st, n2, 1.0 / hn, wj, wj);
    }
    // y = H^-1 g (upper triangular), x += Z y
    for (int i = j - 1; i >= 0; --i) {
      double s = g[i];
      for (int k = i + 1; k < j; ++k) s -= H[(size_t)i * m + k] * y[k];
      y[i] = s / H[(size_t)i * m + i];
    }
    for (int i = 0; i < j; ++i) h->h_small[i] = y[i];
    HIPCHK(hipMemcpyAsync(h->d_small, h->h_small, sizeof(double) * j, hipMemcpyHostToDevice, h->st));
    pgxk_lincomb(h->st, n2, j, h->Z, n2, h->d_small, x, 1);
    HIPCHK(hipStreamSynchronize(h->st));
    // the loop head recomputes the TRUE residual b - Jx: it decides convergence, not the Arnoldi estimate
  }
  *its_out = its;
  *relres = res / bnorm;
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI: fine-grained calls
// ------------------------------------------------------------------------------------------------
extern "C" int pgx_residual(pgx_handle* h, const double* x, double* F, double* fnorm) {
  NEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = copy_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  residual_dev(h, xd, h->F);
  if (fnorm) {
    int rc = dev_norm(h, h->F, fnorm);
    if (rc) return rc;
  }
  if (F) return copy_out(h, F, h->F);
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}

extern "C" int pgx_jacobian_fill(pgx_handle* h, const double* x) {
  NEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = copy_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  jacobian_dev(h, xd);
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}

extern "C" int pgx_csr_export(pgx_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* K,
                              double* M, double* D) {
  NEED(h);
  if (nrows) *nrows = h->n;
  if (nnz) *nnz = h->nnz;
  if (rowptr) memcpy(rowptr, h->h_rowptr.data(), sizeof(int32_t) * (h->n + 1));
  if (col) memcpy(col, h->h_col.data(), sizeof(int32_t) * h->nnz);
  if (K) HIPCHK(hipMemcpy(K, h->Kv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  if (M) HIPCHK(hipMemcpy(M, h->Mv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  if (D) {
    if (!h->jac_valid) {
      h->err = "pgx_csr_export(D) before pgx_jacobian_fill";
      return PGX_ESTATE;
    }
    HIPCHK(hipMemcpy(D, h->Dv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  }
  return PGX_OK;
}

extern "C" int pgx_spmv(pgx_handle* h, const double* x, double* y) {
  NEED(h);
  if (!x || !y) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_spmv before pgx_jacobian_fill";
    return PGX_ESTATE;
  }
  int rc = copy_in(h, h->V, x);
  if (rc) return rc;
  spmv_dev(h, h->V, h->w);
  return copy_out(h, y, h->w);
}

extern "C" int pgx_spmv_bench(pgx_handle* h, int reps, double* avg_ms, double* bytes) {
  NEED(h);
  if (reps < 1 || !avg_ms) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_spmv_bench before pgx_jacobian_fill";
    return PGX_ESTATE;
  }
  const size_t n2 = 2 * (size_t)h->n;
  pgxk_set(h->st, n2, 1.0, h->V);
  for (int k = 0; k < 3; ++k) spmv_dev(h, h->V, h->w);
  HIPCHK(hipEventRecord(h->e0, h->st));
  for (int k = 0; k < reps; ++k) spmv_dev(h, h->V, h->w);
  HIPCHK(hipEventRecord(h->e1, h->st));
  HIPCHK(hipEventSynchronize(h->e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->e0, h->e1));
  *avg_ms = (double)ms / reps;
  if (bytes)  // one pattern (4 B) + three value streams (24 B) per scalar nnz; rowptr; x read once; y written
    *bytes = 28.0 * h->nnz + 4.0 * (h->n + 1) + 8.0 * n2 + 8.0 * n2;
  return PGX_OK;
}

extern "C" int pgx_observables(pgx_handle* h, double out[6]) {
  NEED(h);
  if (!out) return PGX_EINVAL;
  {
    PhaseTimer t(h, 6);
    pgxk_observables(h->st, h->nc, h->n, h->cells, h->coords, h->x, h->xk, h->alpha, h->f, h->q, h->obs_partials,
                     h->obs_blocks, h->d_out6);
  }
  HIPCHK(hipMemcpyAsync(h->h_small, h->d_out6, sizeof(double) * 6, hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  for (int k = 0; k < 6; ++k) out[k] = h->h_small[k];
  return PGX_OK;
}

extern "C" int pgx_profile_enable(pgx_handle* h, int on) {
  if (!h) return PGX_EINVAL;
  h->prof = on != 0;
  return PGX_OK;
}
extern "C" int pgx_profile_get(pgx_handle* h, double ms[8], int reset) {
  if (!h || !ms) return PGX_EINVAL;
  for (int k = 0; k < 8; ++k) ms[k] = h->ms[k];
  if (reset)
    for (int k = 0; k < 8; ++k) h->ms[k] = 0.0;
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// Newton driver: SNES newtonls + linesearch none (full step) convergence logic, SURVEY.md App. A.4;
// options of obstacle_pg.py:128-139; "copy back only if converged" of lvpp/problem.py:121-123.
// ------------------------------------------------------------------------------------------------
extern "C" int pgx_newton_solve(pgx_handle* h, const pgx_snes_opts* opts, int* reason, int* its_out, int* lin_out) {
  NEED(h);
  if (!opts || !reason) return PGX_EINVAL;
  const size_t n2 = 2 * (size_t)h->n;
  hipEvent_t w0 = nullptr, w1 = nullptr;
  if (h->prof) {
    hipEventCreate(&w0);
    hipEventCreate(&w1);
    hipEventRecord(w0, h->st);
  }
  int its = 0, lin = 0, rsn = 0;
  double fnorm = 0, fnorm0 = 0, ttol = 0;
  int rc = PGX_OK;
  HIPCHK(hipMemcpyAsync(h->xw, h->x, n2 * sizeof(double), hipMemcpyDeviceToDevice, h->st));
  residual_dev(h, h->xw, h->F);
  rc = dev_norm(h, h->F, &fnorm);
  if (rc) return rc;
  fnorm0 = fnorm;
  if (opts->monitor) printf("  0 SNES Function norm %.12e\n", fnorm);
  if (!std::isfinite(fnorm))
    rsn = PGX_SNES_DIVERGED_FNORM_NAN;
  else if (fnorm < opts->snes_atol)
    rsn = PGX_SNES_CONVERGED_FNORM_ABS;
  ttol = fnorm * opts->snes_rtol;
  while (rsn == 0) {
    if (its >= opts->snes_max_it) {
      rsn = PGX_SNES_DIVERGED_MAX_IT;
      break;
    }
    jacobian_dev(h, h->xw);
    pgxk_scale_copy(h->st, n2, -1.0, h->F, h->rhs);
    int kits = 0;
    double relres = 0;
    rc = fgmres(h, h->rhs, h->dx, opts, &kits, &relres);
    if (rc) return rc;
    lin += kits;
    ++its;
    if (opts->monitor) printf("    KSP its %d  rel residual %.3e\n", kits, relres);
    if (!(relres <= std::max(opts->ksp_rtol, 1e-7)) || !std::isfinite(relres)) {
      rsn = PGX_SNES_DIVERGED_LINEAR_SOLVE;
      break;
    }
    pgxk_axpy(h->st, n2, 1.0, h->dx, h->xw);
    residual_dev(h, h->xw, h->F);
    rc = dev_norm(h, h->F, &fnorm);
    if (rc) return rc;
    if (opts->monitor) printf("  %d SNES Function norm %.12e\n", its, fnorm);
    if (!std::isfinite(fnorm)) {
      rsn = PGX_SNES_DIVERGED_FNORM_NAN;
    } else if (fnorm < opts->snes_atol) {
      rsn = PGX_SNES_CONVERGED_FNORM_ABS;
    } else if (fnorm <= ttol) {
      rsn = PGX_SNES_CONVERGED_FNORM_RELATIVE;
    } else {
      double snorm, xnorm;
      rc = dev_norm(h, h->dx, &snorm);
      if (rc) return rc;
      rc = dev_norm(h, h->xw, &xnorm);
      if (rc) return rc;
      if (snorm < opts->snes_stol * xnorm)
        rsn = PGX_SNES_CONVERGED_SNORM_RELATIVE;
      else if (fnorm > opts->snes_divtol * fnorm0)
        rsn = PGX_SNES_DIVERGED_DTOL;
    }
  }
  if (rsn > 0) HIPCHK(hipMemcpyAsync(h->x, h->xw, n2 * sizeof(double), hipMemcpyDeviceToDevice, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  if (h->prof) {
    hipEventRecord(w1, h->st);
    hipEventSynchronize(w1);
    float ms = 0;
    hipEventElapsedTime(&ms, w0, w1);
    h->ms[7] += ms;
    hipEventDestroy(w0);
    hipEventDestroy(w1);
  }
  *reason = rsn;
  if (its_out) *its_out = its;
  if (lin_out) *lin_out = lin;
  return PGX_OK;
}
