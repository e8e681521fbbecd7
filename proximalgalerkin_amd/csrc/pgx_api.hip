// Host side of libpgx.so: plan building (CSR pattern, inverted vertex->cell lists), multigrid
// hierarchy, FGMRES, SNES-mirroring Newton driver, and the extern "C" ABI of include/pgx.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pgx.h"
#include "../../include/pgx_nd.h"
#include "pgx_comm.h"
#include "pgx_internal.h"

static thread_local std::string g_create_error;

// Sharded path (include/pgx.h): geometry of this rank's strip on each distributed multigrid level.  Local vertex rows
// of level l are [ghost-low: glo | owned: H | ghost-high: ghi]; g = 2^(ldist-l) is the ghost depth: glo = g (0 on rank 0),
// ghi = g+1 (0 on the last rank, whose H includes the top row of the mesh).  Local row 0 is an even global row on every
// level, so vertex coarsening of the local grid IS the local part of the global coarsening.
struct DistLevel {
  int g, glo, H, ghi;
};
struct Dist {
  bool on = false;
  pgx_comm* comm = nullptr;
  int rank = 0, size = 1, ldist = 0, global_ny = 0;
  std::vector<DistLevel> L;
  GridLevel view{};  // this rank's rows of the first replicated level (pointers into that level's global arrays)
  int view_row0 = 0, view_glo = 0, view_H = 0, view_own0 = 0;
  long long n_halo = 0, n_allreduce = 0, n_vcycle = 0, n_krylov = 0;  // collective counts (pgx_comm_counts)
  double* view_S = nullptr;  // [7 * view.n] strip-shaped coarse stencils before they are merged into the global level
  size_t own_off = 0, own_cnt = 0;  // owned entries per field on level 0: [own_off, own_off + own_cnt)
  // P2 on strips: the edge dofs follow the vertex dofs of a field, ordered by their LOWER vertex (row-major), so the edges whose
  // lower vertex lies in one vertex row form one contiguous block of eb = 3 nx + 1 dofs (nx in the last local row).  An edge is
  // owned by the rank that owns its lower vertex: owned edges = [eown_off, eown_off + eown_cnt) of the edge part.
  size_t eb = 0, eown_off = 0, eown_cnt = 0;
  size_t nk_field() const { return own_cnt + eown_cnt; }  // owned dofs per field (Krylov vectors are owned-compact)
  int cell0 = 0, ncell_own = 0;     // owned cells (row-major cell order)
  double *sb = nullptr, *wc = nullptr;  // local scatter buffer (2n), owned-compact scratch (2*own_cnt)
};

struct pgx_handle {
  int device = 0;
  Dist dist;
  hipStream_t st = nullptr;
  std::string err;
  int n = 0, nc = 0, nx = 0, ny = 0, nnz = 0;  // n = mesh vertices = dofs per field of the P1 space
  int degree = 1;
  int nd = 0;  // dofs per field of the SOLUTION space: n (P1) or n + n_edges (P2)
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
  double* geoq = nullptr;  // order-2 geometry (pgx_create_curved): [cell][quadrature point][5] = |det J|, J^-1; nullptr = affine cells
  float2* zf_out = nullptr;  // FGMRES: the level-0 single-precision cycle leaves its result HERE as float2 (no fp64 copy); see fgmres
  int z_f32 = 1;             // PGX_Z_F32=0: the Z_j of the Krylov method in fp64 (A/B)
  bool dv_lean = false;  // the interior rows of the CSR D values are stale (residual_dev(with_d = 2)); a full fill clears it
  // operator of the solution space (aliases the P1 arrays above for degree 1)
  int32_t *s_rowptr = nullptr, *s_colm = nullptr;
  double *s_K = nullptr, *s_M = nullptr, *s_D = nullptr;
  int s_nnz = 0;
  size_t s_fill_lds = 0;
  int32_t* s_blk = nullptr;  // P2: row blocks of the nnz-balanced stream kernel (k_bspmv_bal): s_nblk + 1 pairs (first row, its rowptr)
  int s_nblk = 0, spmv_bal = 1;
  uint8_t* s_code = nullptr;  // P2 on a uniform mesh: per entry the index of its (K, M) pair in s_tab (k_bspmv_bal<true>)
  double* s_tab = nullptr;    // 256 (K, M) pairs
  int s_ntab = 0, spmv_dict = 1;
  std::vector<int32_t> s_h_rowptr, s_h_col;
  // P2 extras: cell dofs, inverted lists of the P2 plan, P1<->P2 transfers, two-level cycle scratch
  QuadTab2 q2{};
  int32_t *cdofs = nullptr, *p2_v2c_ptr = nullptr, *p2_v2c_ent = nullptr, *p2_v2c_pos = nullptr;
  int32_t *v2e_ptr = nullptr, *v2e = nullptr, *edge_ends = nullptr;
  double *p2_xu = nullptr, *p2_xp = nullptr, *p2_ru = nullptr, *p2_rp = nullptr;
  double* p2_stash = nullptr;  // [16 * nc] element residual vectors of the P2 assembly (deterministic scatter, pgx_p2.hip)
  double *c1_bu = nullptr, *c1_bp = nullptr, *c1_xu = nullptr, *c1_xp = nullptr;
  // vertex-star patch smoother of the P2 level (pgx_patch.hip): NN = slots per patch (0: vertex degree > 7, smoother unavailable)
  int patch_nn = 0, p2_patch = 1, patch_nu = 2;
  int p2_fallback_its = 60, p2_fallbacks = 0;  // patch cycle -> sparse LU after this many Krylov iterations without convergence
  double patch_omega = 0.8;
  bool patch_fresh = false;  // pinv holds the inverses of the current Jacobian
  std::vector<int32_t> patch_dof_host;
  int32_t *pdof = nullptr, *ppos = nullptr;
  void* pinv = nullptr;  // patch inverses: float by default (a smoother inside FGMRES: same Krylov counts as double at 512^2 ... 2048^2,
                         // half the bytes of the stream that bounds the sweep), double with the tuning key PGX_P2_PATCH_F32=0
  int patch_f32 = 1;
  // structured P2 operator apply (pgx_p2st.hip): interior groups [i0, i0 + ni) x [j0, j0 + nj) through the table-driven kernel,
  // the frame rows through the CSR form.  state: 0 off, 1 pattern verified (K / M constants not yet fetched), 2 ready
  struct P2St {
    int state = 0, enable = 1, dist_enable = 1, select = 1, i0 = 0, ni = 0, j0 = 0, nj = 0, nframe = 0, ref_rows[4] = {0, 0, 0, 0};
    P2StTab tab;
    double K[46];
    double* Dst = nullptr;
    float* Dstf = nullptr;
    int32_t* frame = nullptr;
    bool fresh = false;
  } p2st;
  float* s_Df = nullptr;  // float copy of D(psi) for the residuals inside the P2 cycle (k_bspmv_bal<., true>); PGX_P2_RESID_F32=0: fp64
  int p2_resid_f32 = 1;
  int patch_sym = 1;     // float inverses in symmetric packing (pgx_patch.hip: 512 instead of 896 B per patch); PGX_P2_PATCH_SYM=0: full rows
  double *p2_su = nullptr, *p2_sp = nullptr;
  // state
  double *x = nullptr, *xk = nullptr, *F = nullptr, *dx = nullptr, *xw = nullptr, *rhs = nullptr;
  // Krylov workspace
  int restart = 0;
  double *V = nullptr, *Z = nullptr, *w = nullptr, *d_small = nullptr, *partials = nullptr, *partials2 = nullptr;
  double* h_small = nullptr;  // pinned, mapped: the device publishes small results into it (fetch_small), the host polls
  double* h_small_dev = nullptr;              // device view of h_small
  unsigned long long* h_seq = nullptr;        // sequence word behind the payload (host view / device view)
  unsigned long long* h_seq_dev = nullptr;
  unsigned long long seq = 0;
  int lean_d = 1;           // PGX_LEAN_D=0: the Newton loop always writes the CSR form of D(psi) as well
  int spmv_d4 = 1;          // PGX_SPMV_D4=0: the matrix-free operator apply reads the four D arrays instead of its double4 copy
  int lazy_norm = 1;        // PGX_LAZY_NORM=0: every new Krylov vector is normalised in place (one more pass over it per iteration)
  double rhs_scale = 1.0;   // factor the next level-0 V-cycle applies to its fp64 right-hand side as it reads it (lazy normalisation)
  int host_poll = 1;  // PGX_HOST_POLL=0: hipMemcpyAsync + hipStreamSynchronize for every small read-back (round 3)
  // multigrid
  std::vector<GridLevel> lev;
  double *tmp_u = nullptr, *tmp_p = nullptr, *res_u = nullptr, *res_p = nullptr;  // level-0 scratch (each n)
  double omega_now = 0.0;  // smoother damping in force for the current Newton solve (fgmres lowers it on stagnation)
  int coarse_sweeps = 4;  // prototype (oracle/krylov_proto.py): 2..60 sweeps give identical Krylov counts
  int tail_start = -1;  // first level handled by the fused k_mg_tail launch (-1: none)
  int xcd_remap = 2;    // bit 1 of the `first` kernel argument; PGX_XCD_REMAP=0 disables (A/B: +1..3 %)
  int tail_verts = 1100;
  int spmv_stream = 1;  // PGX_SPMV_STREAM=0: 8-lanes-per-row kernel instead of the CSR-stream kernel
  int nu_coarse = 0;    // PGX_NU_COARSE: cap on the sweeps of unfused (small) levels; 0 = same as the fine levels
  int fused_k3 = 1;     // PGX_FUSED_K3=0: two sweeps per launch even when nu is a multiple of 3
  int fused_legs = 1;   // PGX_FUSED_LEGS=0: one launch per sweep / residual / restriction / prolongation
  int fused_min = 4000;  // fused legs from 65^2 vertices up (measured with the row-mapped kernels: 2048^2 460 -> 451 ms, 1024^2
                         // 113.5 -> 106.4, 512^2 68.5 -> 61.5 against 60000; the first, tile-mapped kernels needed >= 60000)
  int cgs_selective = 1;  // second Gram-Schmidt projection only when the first one cancelled (PGX_CGS_SELECTIVE=0: always)
  long cgs_skipped = 0, cgs_total = 0;
  // second projection iff |w'|^2 < eta^2 |w|^2.  The textbook eta = 1/sqrt(2) re-orthogonalises 249 of 266 iterations at 2048^2 (a
  // good preconditioner makes w = J M^-1 v_j ~ v_j: the first projection always cancels most of w); eta = 0.01 bounds the loss of
  // orthogonality per step by 100 eps, re-orthogonalises 11 of 266 and leaves every Krylov count unchanged: 448 -> 416 ms per solve
  double cgs_eta2 = 1e-4;
  TailArgs tail{};
  // sparse direct preconditioner (pc_type lu): nested-dissection multifrontal LU of the mixed Newton matrix (pgx_nd.hip)
  pgx_nd* lu = nullptr;
  // pgx_create_lu_dist: this handle is one of `size` REPLICAS (whole mesh, whole iterate on every rank) whose sparse LU is
  // distributed over the ranks (pgx_nd_create_dist).  Residuals and observables are broadcast from rank 0 so that the
  // replicas stay bitwise identical (the P2 assembly uses atomics) and issue the same collectives.
  pgx_comm* lu_comm = nullptr;
  double* Jmix = nullptr;  // [4 * s_nnz] values of the mixed CSR matrix in the layout lu was created with
  bool dh_interior = false;
  int stag_its = 12;       // PGX_STAG_ITS / PGX_STAG_GAIN: stagnation test of the smoother damping (fgmres)
  double stag_gain = 1e-4;
  int smooth_d32 = 0;      // PGX_SMOOTH_D32=1: the finest-level smoother reads a single-precision copy of the D stencils.  OFF by
                           // default: its launches get 10 % shorter, but the operator apply that follows the V-cycle no longer finds
                           // the double stencils in the Infinity Cache (k_st_spmv_r 0.74 -> 0.55 of the HBM peak) - net +0.9 %
  int k6_max = 0;          // levels with at most this many vertices run 6 sweeps per smoother launch (PGX_K6_MAX)
  int mg_f32 = 1;          // PGX_MG_F32=0: the round-3 fp64 V-cycle.  Default: single-precision V-cycle legs (pgx_mg32.hip) on every
                           // uniform level with at least f32_min vertices above the fused tail - half the bytes per launch
  int f32_min = 4000;      // PGX_F32_MIN
  int f32_k6_max = 300000;   // PGX_F32_K6_MAX: levels with at most this many vertices (513^2) run six sweeps per launch; us per V-cycle
                             // at 2048^2: off 459.7, up to 129^2 453.5, up to 257^2 448.5, up to 513^2 444.3
  int f32_rr_max = 300000;   // PGX_F32_RR_MAX (measured, us per V-cycle from that level: 257^2 and below -2 each, 513^2 +3, 1025^2 +5, 2049^2 +12): levels with at most this many vertices fuse the residual + restriction into the last
                             // pre-smoothing launch (pgx_mg32.hip, RR mode)
  int resid_grid = 1;      // uniform structured P1: residual + D(psi) through k_resid_fill_grid (PGX_RESID_GRID=0: general kernel)
  int spmv_stencil = 1;    // structured P1: the outer-Krylov operator apply through the level-0 stencil kernels (matrix-free)
  bool lu_active = false;  // the current Newton solve preconditions with the factorisation
  bool check_replicas = false;  // PGX_CHECK_REPLICAS=1: assert that the replicas' residuals are bitwise identical
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
    static const char* const names[8] = {"pgx:residual", "pgx:jacobian", "pgx:mg_setup/lu_factor", "pgx:spmv", "pgx:precond",
                                         "pgx:orthogonalise", "pgx:observables", "pgx:newton_solve"};
    pgx_roctx(names[s & 7]);
    if (h->prof) hipEventRecord(h->e0, h->st);
  }
  ~PhaseTimer() {
    pgx_roctx(nullptr);
    if (h->prof) {
      hipEventRecord(h->e1, h->st);
      hipEventSynchronize(h->e1);
      float ms = 0;
      hipEventElapsedTime(&ms, h->e0, h->e1);
      h->ms[slot] += ms;
    }
  }
};

// ---- tuning table (pgx_scope.h: pgx_tune) ----
namespace {
std::mutex g_tune_mu;
std::atomic<int> g_tune_gen{0};
// values are interned in a pool that is never freed: pgx_tune hands out pointers that outlive any later pgx_tuning_set
std::map<std::string, const std::string*>& tune_table() {
  static std::map<std::string, const std::string*> t;
  return t;
}
std::deque<std::string>& tune_pool() {
  static std::deque<std::string> p;
  return p;
}
}  // namespace
const char* pgx_tune(const char* name) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  auto& t = tune_table();
  auto it = t.find(name);
  return it == t.end() ? nullptr : it->second->c_str();
}
int pgx_tune_gen() { return g_tune_gen.load(std::memory_order_acquire); }
extern "C" int pgx_tuning_set(const char* key, const char* value) {
  if (!key || strncmp(key, "PGX_", 4) != 0) return PGX_EINVAL;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  if (value) {
    const std::string* interned = nullptr;
    for (const std::string& s : tune_pool())
      if (s == value) interned = &s;
    if (!interned) {
      tune_pool().emplace_back(value);  // std::deque never moves its elements
      interned = &tune_pool().back();
    }
    tune_table()[key] = interned;
  } else {
    tune_table().erase(key);
  }
  g_tune_gen.fetch_add(1, std::memory_order_release);
  return PGX_OK;
}

extern "C" void pgx_default_opts(pgx_snes_opts* o) {
  o->snes_rtol = 1e-8;
  o->snes_atol = 1e-50;
  o->snes_stol = 1e-8;
  o->snes_divtol = 1e4;
  o->snes_max_it = 50;
  o->ksp_rtol = 0.0;  // 0 = auto: 1e-10 for P1, 1e-11 for P2 (measured effect on the final u: DESIGN.md section 3)
  o->ksp_max_it = 200;
  o->ksp_restart = 30;
  o->mg_nu = 6;  // 2048^2 sweep (profiles/): nu=2 780 ms, 4 707, 6 660, 8 692, 10 763 per solve - more smoothing shrinks the
                  // Krylov space, whose orthogonalisation cost grows with its square
  // damping, re-measured in round 2 on the row-mapped kernels (2048^2, ms per solve / Krylov iterations per solve): 0.60 338, 0.65 324,
  // 0.70 314 / -, 0.72 311 / 252, 0.75 302 / 246, 0.76 - / 248, 0.77 - / 250, 0.78 332 / 265, 0.80 334 / 267, 0.85 359, 0.90 374: the
  // optimum of the smoothing factor is flat between 0.72 and 0.77; from 0.78 on one late Newton step trips the stagnation
  // test and finishes at omega_safe.  1024^2 and 512^2 do not care (87-89 ms, 58 ms for 0.70-0.80).  Round 1 used 0.8.
  o->mg_omega = 0.75;
  o->monitor = 0;
  o->pc_type = 0;  // auto: multigrid for P1, sparse LU for P2
  o->linesearch = 0;
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

// P2 plan: scalar CSR pattern over [vertex | edge] dofs, dof -> (cell, local dof) lists with the row positions
// of the cell's 6 dofs (two int32 per entry), vertex -> edge-dof lists and edge end points for the transfers.
static int build_plan_p2(pgx_handle* h, const pgx_mesh* m, const std::vector<uint8_t>& hmask) {
  const int n = h->nd, nc = m->n_cells, nv = m->n_vertices;
  const int32_t* cd = m->cell_dofs;
  std::vector<int32_t> vptr(n + 1, 0);
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 6; ++a) {
      const int v = cd[6 * c + a];
      if (v < 0 || v >= n || (a < 3 && v != m->cells[3 * c + a]) || (a >= 3 && v < nv)) {
        h->err = "cell_dofs must be [vertex ids (== cells) | edge dofs >= n_vertices]";
        return PGX_EINVAL;
      }
      vptr[v + 1]++;
    }
  for (int i = 0; i < n; ++i) vptr[i + 1] += vptr[i];
  std::vector<int32_t> vent(vptr[n]), fillp(vptr.begin(), vptr.end() - 1);
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 6; ++a) vent[fillp[cd[6 * c + a]]++] = c * 8 + a;
  std::vector<int32_t>& rowptr = h->s_h_rowptr;
  std::vector<int32_t>& col = h->s_h_col;
  rowptr.assign(n + 1, 0);
  col.clear();
  col.reserve((size_t)n * 12);
  std::vector<int32_t> tmp;
  for (int i = 0; i < n; ++i) {
    tmp.clear();
    for (int k = vptr[i]; k < vptr[i + 1]; ++k) {
      const int c = vent[k] >> 3;
      for (int b = 0; b < 6; ++b) tmp.push_back(cd[6 * c + b]);
    }
    if (tmp.empty()) tmp.push_back(i);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if (tmp.size() > 255) {
      h->err = "dof degree > 255 unsupported";
      return PGX_EINVAL;
    }
    col.insert(col.end(), tmp.begin(), tmp.end());
    rowptr[i + 1] = (int32_t)col.size();
  }
  if (col.size() > 0x7fffffffu) {
    h->err = "nnz overflows int32";
    return PGX_EINVAL;
  }
  h->s_nnz = (int)col.size();
  std::vector<int32_t> vpos(2 * vent.size());
  for (int i = 0; i < n; ++i) {
    const int32_t* rb = col.data() + rowptr[i];
    const int32_t* re = col.data() + rowptr[i + 1];
    for (int k = vptr[i]; k < vptr[i + 1]; ++k) {
      const int c = vent[k] >> 3;
      uint32_t p0 = 0, p1 = 0;
      for (int b = 0; b < 6; ++b) {
        const uint32_t pp = (uint32_t)(std::lower_bound(rb, re, cd[6 * c + b]) - rb);
        if (b < 4) p0 |= pp << (8 * b);
        else p1 |= pp << (8 * (b - 4));
      }
      vpos[2 * k] = (int32_t)p0;
      vpos[2 * k + 1] = (int32_t)p1;
    }
  }
  size_t maxlen = 0;
  for (int i0 = 0; i0 < n; i0 += PGX_BLOCK) {
    const int i1 = std::min(i0 + PGX_BLOCK, n);
    maxlen = std::max(maxlen, (size_t)(rowptr[i1] - rowptr[i0]));
  }
  std::vector<int32_t> blk{0};
  for (int r = 0; r < n;) {  // greedy: as many rows as fit PGX_BAL_CAP entries (a P2 row has at most 23), at most PGX_BLOCK rows
    int e = r;
    while (e < n && e - r < PGX_BLOCK && rowptr[e + 1] - rowptr[r] <= PGX_BAL_CAP) ++e;
    if (e == r) {
      blk.clear();  // a single row beyond the capacity: keep the unbalanced kernel
      break;
    }
    blk.push_back(e);
    r = e;
  }
  h->s_fill_lds = maxlen * sizeof(double);
  if (h->s_fill_lds > 60 * 1024) {
    h->err = "row block too dense for the LDS-staged fill";
    return PGX_EINVAL;
  }
  // transfers: edge end points (local edge i is opposite local vertex i) and vertex -> edge dofs
  const int ne = n - nv;
  std::vector<int32_t> ends(2 * (size_t)ne, -1), eptr(nv + 1, 0);
  for (int c = 0; c < nc; ++c)
    for (int i = 0; i < 3; ++i) {
      const int e = cd[6 * c + 3 + i] - nv;
      const int a = m->cells[3 * c + (i + 1) % 3], b = m->cells[3 * c + (i + 2) % 3];
      const int lo = std::min(a, b), hi = std::max(a, b);
      if (ends[2 * e] >= 0 && (ends[2 * e] != lo || ends[2 * e + 1] != hi)) {
        h->err = "cell_dofs: an edge dof is attached to two different vertex pairs";
        return PGX_EINVAL;
      }
      ends[2 * e] = lo;
      ends[2 * e + 1] = hi;
    }
  for (int e = 0; e < ne; ++e) {
    if (ends[2 * e] < 0) {
      h->err = "cell_dofs: unused edge dof";
      return PGX_EINVAL;
    }
    eptr[ends[2 * e] + 1]++;
    eptr[ends[2 * e + 1] + 1]++;
  }
  if (h->dist.on) {  // sharded P2 addresses the edges of a vertex row as one contiguous block (Dist::eb)
    const int sx = h->nx + 1;
    std::vector<int> per_row(h->ny + 1, 0);
    bool sorted = true;
    for (int e = 0; e < ne; ++e) {
      if (e && ends[2 * e] < ends[2 * (e - 1)]) sorted = false;
      per_row[ends[2 * e] / sx]++;
    }
    for (int j = 0; j <= h->ny && sorted; ++j) sorted = per_row[j] == (j < h->ny ? 3 * h->nx + 1 : h->nx);
    if (!sorted) {
      h->err = "pgx_create_sharded (P2): edge dofs must be numbered by their lower vertex, row by row (3 nx + 1 per vertex row)";
      return PGX_EINVAL;
    }
  }
  for (int i = 0; i < nv; ++i) eptr[i + 1] += eptr[i];
  std::vector<int32_t> elist(eptr[nv]), ef(eptr.begin(), eptr.end() - 1);
  for (int e = 0; e < ne; ++e) {
    elist[ef[ends[2 * e]]++] = nv + e;
    elist[ef[ends[2 * e + 1]]++] = nv + e;
  }
  std::vector<int32_t> colm(col.size());
  for (size_t k = 0; k < col.size(); ++k) colm[k] = col[k] | (hmask[col[k]] ? (int32_t)0x80000000 : 0);
  // structured operator apply (pgx_p2st.hip): derive the 46-entry table from one interior group, verify it on every group of the
  // largest rectangle of groups that match, list the rows of the frame around it
  if (h->structured && (!h->dist.on || h->p2st.dist_enable) && h->p2st.enable && h->nx >= 8 && h->ny >= 8) {
    const int nx = h->nx, ny = h->ny, sx = nx + 1;
    auto ebase = [&](int i, int j) { return nv + j * (3 * nx + 1) + 3 * i; };
    bool ok = (nv == sx * (ny + 1)) && (ne == ny * (3 * nx + 1) + nx);
    for (int j = 0; j < ny && ok; ++j)
      for (int i = 0; i < nx && ok; ++i) {  // edges numbered by lower vertex: H, V, D of vertex (i, j)
        const int v = j * sx + i, e = ebase(i, j) - nv;
        ok = ends[2 * e] == v && ends[2 * e + 1] == v + 1 && ends[2 * e + 2] == v && ends[2 * e + 3] == v + sx &&
             ends[2 * e + 4] == v && ends[2 * e + 5] == v + sx + 1;
      }
    pgx_handle::P2St& S = h->p2st;
    const int ir = nx / 2, jr = ny / 2;  // reference group
    int nt[4] = {19, 9, 9, 9}, off[4] = {0, 19, 28, 37};
    if (ok) {
      const int vr = jr * sx + ir, er = ebase(ir, jr);
      for (int t = 0; t < 4 && ok; ++t) {
        const int row = t == 0 ? vr : er + t - 1;
        S.ref_rows[t] = row;
        ok = rowptr[row + 1] - rowptr[row] == nt[t];
        for (int k = 0; k < nt[t] && ok; ++k) {
          const int c = col[rowptr[row] + k];
          S.tab.isedge[off[t] + k] = c >= nv;
          S.tab.delta[off[t] + k] = c >= nv ? c - er : c - vr;
        }
      }
    }
    auto group_ok = [&](int i, int j) -> bool {
      if (i < 1 || j < 1 || i >= nx || j >= ny) return false;
      const int v = j * sx + i, eb = ebase(i, j);
      for (int t = 0; t < 4; ++t) {
        const int row = t == 0 ? v : eb + t - 1;
        if (hmask[row] || rowptr[row + 1] - rowptr[row] != nt[t]) return false;
        for (int k = 0; k < nt[t]; ++k) {
          const int c = col[rowptr[row] + k];
          if (hmask[c] || c != (S.tab.isedge[off[t] + k] ? eb : v) + S.tab.delta[off[t] + k]) return false;
        }
      }
      return true;
    };
    if (ok && group_ok(ir, jr)) {
      int i0 = ir, i1 = ir, j0 = jr, j1 = jr;
      while (group_ok(i0 - 1, jr)) --i0;
      while (group_ok(i1 + 1, jr)) ++i1;
      while (group_ok(ir, j0 - 1)) --j0;
      while (group_ok(ir, j1 + 1)) ++j1;
      for (int j = j0; j <= j1 && ok; ++j)
        for (int i = i0; i <= i1 && ok; ++i) ok = group_ok(i, j);
      if (ok) {
        std::vector<int32_t> frame;
        auto fast = [&](int v) {
          const int i = v % sx, j = v / sx;
          return i >= i0 && i <= i1 && j >= j0 && j <= j1;
        };
        for (int v = 0; v < nv; ++v)
          if (!fast(v)) frame.push_back(v);
        for (int e = 0; e < ne; ++e)
          if (!fast(ends[2 * e])) frame.push_back(nv + e);
        // LDS form: decode every entry into (dj, di[, kind]) and its offset in the block tiles of k_p2st_apply_lds
        {
          const int VW = PGX_P2ST_BW + 2, EW = 3 * VW, erow = 3 * nx + 1;
          bool lds = true;
          for (int e = 0; e < 46 && lds; ++e) {
            const int d = S.tab.delta[e];
            if (!S.tab.isedge[e]) {
              const int dj = (int)std::lround((double)d / sx), di = d - dj * sx;
              lds = std::abs(dj) <= 1 && std::abs(di) <= 1;
              S.tab.lofs[e] = (dj + 1) * VW + di + 1;
            } else {
              const int dj = (int)std::lround((double)d / erow), rem = d - dj * erow;  // rem = 3 di + kind, di in {-1, 0, 1}
              const int di = (rem + 3) / 3 - 1, kind = rem - 3 * di;
              lds = std::abs(dj) <= 1 && std::abs(di) <= 1 && kind >= 0 && kind <= 2;
              S.tab.lofs[e] = (dj + 1) * EW + 3 * (di + 1) + kind;
            }
          }
          S.tab.lds = lds ? 1 : 0;
          if (const char* e = pgx_tune("PGX_P2ST_LDS")) S.tab.lds = S.tab.lds && atoi(e);
        }
        S.i0 = i0, S.ni = i1 - i0 + 1, S.j0 = j0, S.nj = j1 - j0 + 1, S.nframe = (int)frame.size();
        DALLOC(S.frame, std::max<size_t>(frame.size(), 1));
        HIPCHK(hipMemcpy(S.frame, frame.data(), sizeof(int32_t) * frame.size(), hipMemcpyHostToDevice));
        S.state = 1;
      }
    }
  }
  DALLOC(h->s_rowptr, n + 1);
  {  // vertex-star patches: slot 0 = the vertex, then the edge dofs that meet in it (pgx_patch.hip)
    int maxdeg = 0;
    for (int v = 0; v < nv; ++v) maxdeg = std::max(maxdeg, eptr[v + 1] - eptr[v]);
    h->patch_nn = maxdeg + 1 <= 7 ? 7 : (maxdeg + 1 <= 8 ? 8 : 0);
    if (h->patch_nn) {
      const int NN = h->patch_nn;
      h->patch_dof_host.assign((size_t)nv * NN, -1);
      for (int v = 0; v < nv; ++v) {
        int32_t* d = h->patch_dof_host.data() + (size_t)v * NN;
        d[0] = v;
        for (int k = eptr[v]; k < eptr[v + 1]; ++k) d[1 + k - eptr[v]] = elist[k];
      }
    }
  }
  if (blk.size() > 1) {
    h->s_nblk = (int)blk.size() - 1;
    std::vector<int32_t> blk2(2 * blk.size());  // (first row, its rowptr) per block boundary: the kernel's one metadata load
    for (size_t i = 0; i < blk.size(); ++i) {
      blk2[2 * i] = blk[i];
      blk2[2 * i + 1] = rowptr[blk[i]];
    }
    DALLOC(h->s_blk, blk2.size());
    HIPCHK(hipMemcpy(h->s_blk, blk2.data(), sizeof(int32_t) * blk2.size(), hipMemcpyHostToDevice));
  }
  DALLOC(h->s_colm, colm.size());
  DALLOC(h->p2_v2c_ptr, n + 1);
  DALLOC(h->p2_v2c_ent, vent.size());
  DALLOC(h->p2_v2c_pos, vpos.size());
  DALLOC(h->cdofs, (size_t)6 * nc);
  DALLOC(h->edge_ends, ends.size());
  DALLOC(h->v2e_ptr, nv + 1);
  DALLOC(h->v2e, elist.size());
  HIPCHK(hipMemcpy(h->s_rowptr, rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->s_colm, colm.data(), sizeof(int32_t) * colm.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->p2_v2c_ptr, vptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->p2_v2c_ent, vent.data(), sizeof(int32_t) * vent.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->p2_v2c_pos, vpos.data(), sizeof(int32_t) * vpos.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->cdofs, cd, sizeof(int32_t) * 6 * (size_t)nc, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->edge_ends, ends.data(), sizeof(int32_t) * ends.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->v2e_ptr, eptr.data(), sizeof(int32_t) * (nv + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->v2e, elist.data(), sizeof(int32_t) * elist.size(), hipMemcpyHostToDevice));
  return PGX_OK;
}

// Is the stencil the same at every interior vertex (uniform grid)?  Then interior rows take their K and M
// coefficients from kernel arguments.  Setup-time host check on a downloaded copy.
// (K, M) dictionary of the P2 level for k_bspmv_bal<true>: distinct pairs after rounding to a grid of 2^-40 of the largest entry
// (finer than the tolerance of detect_uniform below; a function of each entry alone: deterministic).  Seeded with nothing; every round lists up to 4096 unmatched entries, the host adds their distinct values.
// More than 256 pairs (a non-uniform mesh) => no dictionary, the kernel streams K and M as before.
static int allreduce_dev(pgx_handle* h, double* dev, size_t n);
// max |x_i| through atomicMax on the bit pattern (non-negative doubles order like their 64-bit patterns)
__global__ void __launch_bounds__(256) k_maxabs(int64_t n, const double* __restrict__ x, unsigned long long* out) {
  double m = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmax(m, fabs(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

static int build_km_dictionary(pgx_handle* h) {
  const int64_t nnz = h->s_nnz;
  const int cap = 4096;
  uint8_t* code = nullptr;
  double* tab = nullptr;
  int* fail = nullptr;
  double* fail_v = nullptr;
  DALLOC(code, (size_t)nnz);
  DALLOC(tab, 512);
  DALLOC(fail, 2);
  DALLOC(fail_v, 2 * cap);
  // scale: the largest |K| and |M| over ALL entries (round 3 looked at the first 65536 only - vertex rows - and missed the larger
  // edge-edge entries of P2), and on a sharded handle the largest over all ranks, so that every rank rounds the same
  // global operator to the same grid
  double kmax = 0.0, mmax = 0.0;
  {
    unsigned long long* dmax = nullptr;
    DALLOC(dmax, 2);
    HIPCHK(hipMemsetAsync(dmax, 0, 2 * sizeof(unsigned long long), h->st));
    const int nb = (int)std::min<int64_t>((nnz + 255) / 256, 4096);
    hipLaunchKernelGGL(k_maxabs, dim3(nb), dim3(256), 0, h->st, nnz, h->s_K, dmax);
    hipLaunchKernelGGL(k_maxabs, dim3(nb), dim3(256), 0, h->st, nnz, h->s_M, dmax + 1);
    unsigned long long hm[2];
    HIPCHK(hipMemcpyAsync(hm, dmax, sizeof(hm), hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    memcpy(&kmax, &hm[0], 8);
    memcpy(&mmax, &hm[1], 8);
    if (h->dist.on) {  // max over the ranks through the sum all-reduce: slot `rank` of a zeroed buffer
      const int R = h->dist.size;
      std::vector<double> slots(2 * (size_t)R, 0.0);
      slots[2 * (size_t)h->dist.rank] = kmax;
      slots[2 * (size_t)h->dist.rank + 1] = mmax;
      double* ds = nullptr;
      DALLOC(ds, 2 * (size_t)R);
      HIPCHK(hipMemcpyAsync(ds, slots.data(), slots.size() * sizeof(double), hipMemcpyHostToDevice, h->st));
      const int rc = allreduce_dev(h, ds, 2 * (size_t)R);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(slots.data(), ds, slots.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
      HIPCHK(hipStreamSynchronize(h->st));
      for (int r = 0; r < R; ++r) {
        kmax = std::max(kmax, slots[2 * (size_t)r]);
        mmax = std::max(mmax, slots[2 * (size_t)r + 1]);
      }
    }
  }
  // grid = 2^-40 (9e-13) of the largest entry, a power of two: rounding to it is exact arithmetic, the same on host and device
  const double tk = std::ldexp(1.0, std::ilogb(kmax > 0 ? kmax : 1.0) - 40), tm = std::ldexp(1.0, std::ilogb(mmax > 0 ? mmax : 1.0) - 40);
  std::vector<double> table(512, 0.0);
  int ntab = 0;
  std::vector<double> fv(2 * cap);
  bool ok = false;
  for (int round = 0; round < 64; ++round) {
    HIPCHK(hipMemcpyAsync(tab, table.data(), 512 * sizeof(double), hipMemcpyHostToDevice, h->st));
    HIPCHK(hipMemsetAsync(fail, 0, 2 * sizeof(int), h->st));
    pgxk_dict_assign(h->st, nnz, h->s_K, h->s_M, ntab, tab, tk, tm, code, fail, cap, fail_v);
    int f[2];
    HIPCHK(hipMemcpyAsync(f, fail, sizeof(f), hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    const int nf = std::min(f[0], cap);
    if (nf == 0) {
      ok = true;
      break;
    }
    HIPCHK(hipMemcpy(fv.data(), fail_v, 2 * nf * sizeof(double), hipMemcpyDeviceToHost));
    bool full = false;
    for (int i = 0; i < nf && !full; ++i) {
      const double kv = fv[2 * i], mv = fv[2 * i + 1];  // already rounded to the grid by the kernel
      bool have = false;
      for (int t = 0; t < ntab && !have; ++t) have = kv == table[2 * t] && mv == table[2 * t + 1];
      if (have) continue;
      if (ntab == 256) {
        full = true;
        break;
      }
      table[2 * ntab] = kv;
      table[2 * ntab + 1] = mv;
      ++ntab;
    }
    if (full) break;
  }
  if (pgx_tune("PGX_SPMV_DICT_VERBOSE")) fprintf(stderr, "pgx: (K,M) dictionary: %d pairs, %s\n", ntab, ok ? "in use" : "overflow - not used");
  if (ok) {
    h->s_code = code;
    h->s_tab = tab;
    h->s_ntab = ntab;
  }  // else: the arrays stay allocated (freed with the handle) and unused
  return PGX_OK;
}

static int detect_uniform(pgx_handle* h, GridLevel& L) {
  L.uniform = 0;
  L.interior_free = 0;
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
  std::vector<uint8_t> mk(L.n);
  HIPCHK(hipMemcpyAsync(mk.data(), L.mask, L.n, hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  L.interior_free = 1;
  for (int j = 1; j < L.ny && L.interior_free; ++j)
    for (int i = 1; i < L.nx; ++i)
      if (mk[(size_t)j * sx + i]) {
        L.interior_free = 0;
        break;
      }
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// sharded path: strip partition arithmetic (no GPU needed) and the ghost-row exchange
// ------------------------------------------------------------------------------------------------
// Ghost depth multiplier m (PGX_GHOST_MUL, default 3; every rank must see the same value): distributed level l keeps m 2^(ld-l)
// ghost rows (24 / 12 / 6 for three distributed levels).  An exchange restores validity depth g and a fused smoother launch
// consumes K = 3 of it, so deeper ghosts trade redundant rows (m 2^ld per strip side on the finest level: +19 % of a 256-row
// strip at m = 3) for fewer latency-bound exchanges: 14.2 / 9.2 / 6.2 / 6.2 per Krylov iteration at m = 1 / 2 / 3 / 4
// (pgx_comm_counts, 2048^2 on 4 strips).
static int ghost_mul() {
  static const int m = [] {
    const char* e = pgx_tune("PGX_GHOST_MUL");
    const int v = e ? atoi(e) : 3;
    return v >= 1 && v <= 4 ? v : 3;
  }();
  return m;
}

static int partition_resolve(pgx_partition* pt, std::string& err) {
  if (!pt || pt->size < 1 || pt->rank < 0 || pt->rank >= pt->size || pt->global_ny < 1 || pt->dist_levels < 0) {
    err = "pgx_partition: need 0 <= rank < size, global_ny >= 1, dist_levels >= 0";
    return PGX_EINVAL;
  }
  if (pt->global_ny % pt->size) {
    err = "pgx_partition: global_ny must be divisible by the number of strips";
    return PGX_EINVAL;
  }
  const int Hh = pt->global_ny / pt->size;
  // level l keeps Hh/2^l owned rows and 2^(ld-l) ghost rows: strips must coarsen ld times and still own g+1 = 3 rows
  auto ok = [&](int l) { return l >= 1 && l <= 8 && Hh % (1 << l) == 0 && (Hh >> (l - 1)) >= 4 * ghost_mul(); };
  int ld = pt->dist_levels;
  if (ld == 0) {
    for (ld = 3; ld >= 1 && !ok(ld); --ld) {}
    if (ld < 1) {
      err = "pgx_partition: strips too thin to shard (need global_ny / size divisible by 2 and >= 4)";
      return PGX_EINVAL;
    }
    pt->dist_levels = ld;
  } else if (!ok(ld)) {
    err = "pgx_partition: global_ny / size must be divisible by 2^dist_levels and leave >= 4 rows on the last distributed level";
    return PGX_EINVAL;
  }
  return PGX_OK;
}

extern "C" int pgx_partition_rows(pgx_partition* pt, int32_t* row0, int32_t* nrows, int32_t* own0, int32_t* nown) {
  std::string err;
  const int rc = partition_resolve(pt, err);
  if (rc) {
    g_create_error = err;
    return rc;
  }
  const int Hh = pt->global_ny / pt->size, g0 = ghost_mul() << pt->dist_levels;
  const int r0 = pt->rank * Hh - (pt->rank > 0 ? g0 : 0);
  const int r1 = (pt->rank + 1 < pt->size) ? (pt->rank + 1) * Hh + g0 : pt->global_ny;  // last local vertex row
  if (row0) *row0 = r0;
  if (nrows) *nrows = r1 - r0 + 1;
  if (own0) *own0 = pt->rank * Hh;
  if (nown) *nown = Hh + (pt->rank + 1 == pt->size ? 1 : 0);
  return PGX_OK;
}

// refresh every ghost row of a (u, psi) pair on distributed level l from the owning neighbours
static int halo_level(pgx_handle* h, int l, double* fu, double* fp) {
  const DistLevel& d = h->dist.L[l];
  const size_t sx = (size_t)h->lev[l].nx + 1;
  double* f[2] = {fu, fp};
  const int rc = h->dist.comm->halo(h->st, f, 2, d.glo * sx, (d.g + 1) * sx, 0, d.g * sx, (d.glo + d.H - d.g) * sx,
                                    d.g * sx, (d.glo + d.H) * sx, (d.g + 1) * sx);
  ++h->dist.n_halo;
  if (rc) h->err = h->dist.comm->err;
  return rc;
}
// the same for ONE interleaved (u, psi) float2 field of the single-precision cycle (8 bytes per vertex: one "double" field)
static int halo_level_f(pgx_handle* h, int l, float2* f2) {
  const DistLevel& d = h->dist.L[l];
  const size_t sx = (size_t)h->lev[l].nx + 1;
  double* f[1] = {reinterpret_cast<double*>(f2)};
  const int rc = h->dist.comm->halo(h->st, f, 1, d.glo * sx, (d.g + 1) * sx, 0, d.g * sx, (d.glo + d.H - d.g) * sx, d.g * sx,
                                    (d.glo + d.H) * sx, (d.g + 1) * sx);
  ++h->dist.n_halo;
  if (rc) h->err = h->dist.comm->err;
  return rc;
}
// ghost entries of a SOLUTION-SPACE pair (fu, fp) - each [vertex dofs | edge dofs] for P2 - from their owners
static int halo_solution(pgx_handle* h, double* fu, double* fp) {
  int rc = halo_level(h, 0, fu, fp);
  if (rc || h->degree != 2) return rc;
  // edge part: g full row blocks below and above (the last local row's horizontal edges stay behind: they lie on the outermost,
  // never-valid ghost line)
  const DistLevel& d = h->dist.L[0];
  const size_t eb = h->dist.eb, nv = (size_t)h->n;
  double* f[2] = {fu + nv, fp + nv};
  rc = h->dist.comm->halo(h->st, f, 2, d.glo * eb, d.g * eb, 0, d.g * eb, (size_t)(d.glo + d.H - d.g) * eb, d.g * eb,
                          (size_t)(d.glo + d.H) * eb, d.g * eb);
  ++h->dist.n_halo;
  if (rc) h->err = h->dist.comm->err;
  return rc;
}
static int allreduce_dev(pgx_handle* h, double* dev, size_t n) {
  const int rc = h->dist.comm->allreduce(h->st, dev, n);
  ++h->dist.n_allreduce;
  if (rc) h->err = h->dist.comm->err;
  return rc;
}

// fused tail: every level from `first` on with at most PGX_TAIL_VERTS vertices (and at most PGX_TAIL_MAX of them)
static void setup_tail(pgx_handle* h, int first) {
  const int nl = (int)h->lev.size();
  {  // a coarsest grid that could not be coarsened to a handful of vertices (odd cell counts) gets a sweep
     // count that grows with its size: Jacobi is then a poor but non-trivial coarse solver
    const GridLevel& Lc = h->lev.back();
    if (nl > 1 && Lc.n > 100 && !pgx_tune("PGX_COARSE_SWEEPS")) h->coarse_sweeps = std::min(400, 4 * std::max(Lc.nx, Lc.ny));
  }
  for (int l = std::max(first, 1); l < nl; ++l)
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
      T.interior_free = L.interior_free;
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
}

// Coarsest grid of the hierarchy: coarsening stops at this many cells per side (PGX_MG_MIN_NX).  Round 4: 4 instead of 2 - the 3 x 3
// grid has ONE interior vertex and its visit costs the fused tail 8 us per V-cycle (46.6 -> 38.7 us) for nothing: identical Krylov
// and Newton counts on every test and on the 2048^2 benchmark.
static int mg_min_nx() {
  const char* e = pgx_tune("PGX_MG_MIN_NX");
  return e ? std::max(2, atoi(e)) : 4;
}

// Which levels run the single-precision legs (pgx_mg32.hip): every level with uniform stencils (the row-mapped kernels) and at
// least f32_min vertices above the fused tail, never the coarsest.  fp64 and single-precision levels may alternate: each hands its
// right-hand side down and its correction up in the format of the level that receives it (vcycle / vcycle_f / vcycle_dist_f).
static int setup_f32(pgx_handle* h) {
  if (h->structured && h->degree == 1 && h->spmv_stencil == 1 && h->spmv_d4 && !h->lev.empty() && h->lev[0].uniform && sizeof(dsten_t) == 8)
    DALLOC(h->lev[0].Dd4, h->lev[0].n);  // the operator apply's own copy of D (k_st_spmv_r)
  if (!h->mg_f32 || !h->structured || sizeof(dsten_t) != 8) return PGX_OK;
  const int nl = (int)h->lev.size();
  for (int l = 0; l + 1 < nl; ++l) {
    GridLevel& L = h->lev[l];
    if (!L.uniform || L.n < std::max(h->f32_min, h->fused_min) || (h->tail_start > 0 && l >= h->tail_start)) continue;
    DALLOC(L.Dq, L.n);
    DALLOC(L.xf, L.n);
    DALLOC(L.xf2, L.n);
    DALLOC(L.bf, L.n);
    for (float2* p : {L.xf, L.xf2, L.bf}) HIPCHK(hipMemsetAsync(p, 0, sizeof(float2) * L.n, h->st));
    HIPCHK(hipMemsetAsync(L.Dq, 0, sizeof(float4) * L.n, h->st));
    L.f32 = 1;
  }
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
  if (h->smooth_d32 && sizeof(dsten_t) == 8) DALLOC(h->lev[0].Dh32, (size_t)4 * h->n);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Kv, h->lev[0].K);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Mv, h->lev[0].M);
  int rc = detect_uniform(h, h->lev[0]);
  if (rc) return rc;
  int nx = h->nx, ny = h->ny;
  const int min_nx = mg_min_nx();
  while (nx % 2 == 0 && ny % 2 == 0 && nx > min_nx && ny > min_nx) {
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
  setup_tail(h, 1);
  rc = setup_f32(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}

// all solver-side arrays of one stencil level (K, M are filled by the caller)
static int alloc_level(pgx_handle* h, GridLevel& L, bool contiguous_rhs) {
  DALLOC(L.K, (size_t)7 * L.n);
  DALLOC(L.M, (size_t)7 * L.n);
  DALLOC(L.Dh, (size_t)4 * L.n);
  DALLOC(L.mask, L.n);
  DALLOC(L.xu, L.n);
  DALLOC(L.xp, L.n);
  DALLOC(L.xu2, L.n);
  DALLOC(L.xp2, L.n);
  if (contiguous_rhs) {  // (bu | bp) in one block: one memset + one all-reduce for the pair
    DALLOC(L.bu, (size_t)2 * L.n);
    L.bp = L.bu + L.n;
  } else {
    DALLOC(L.bu, L.n);
    DALLOC(L.bp, L.n);
  }
  DALLOC(L.ru, L.n);
  DALLOC(L.rp, L.n);
  return PGX_OK;
}

// Sharded hierarchy: levels [0, ldist) are this rank's strips (ghost depth 2^(ldist-l)); level ldist and below are the
// GLOBAL grids, replicated on every rank.  Galerkin coarsening of a strip needs no communication: a coarse stencil at
// ghost depth d uses fine stencils up to depth 2d+1, so "valid up to depth g-1" is inherited from level to level.  Only
// the step onto the first replicated level merges the ranks' rows (owned rows, zeros elsewhere, one all-reduce).
static int build_multigrid_dist(pgx_handle* h) {
  Dist& D = h->dist;
  const int ld = D.ldist;
  GridLevel L0{};
  L0.nx = h->nx;
  L0.ny = h->ny;
  L0.n = h->n;
  L0.mask = h->mask;
  h->lev.push_back(L0);
  DALLOC(h->lev[0].K, (size_t)7 * h->n);
  DALLOC(h->lev[0].M, (size_t)7 * h->n);
  DALLOC(h->lev[0].Dh, (size_t)4 * h->n);
  if (h->smooth_d32 && sizeof(dsten_t) == 8) DALLOC(h->lev[0].Dh32, (size_t)4 * h->n);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Kv, h->lev[0].K);
  pgxk_csr_to_stencil(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Mv, h->lev[0].M);
  int rc = detect_uniform(h, h->lev[0]);
  if (rc) return rc;
  for (int l = 1; l < ld; ++l) {
    const GridLevel Fl = h->lev.back();
    GridLevel L{};
    L.nx = Fl.nx / 2;
    L.ny = Fl.ny / 2;
    L.n = (L.nx + 1) * (L.ny + 1);
    rc = alloc_level(h, L, false);
    if (rc) return rc;
    pgxk_coarse_mask(h->st, L, L.mask, Fl);
    pgxk_rap7(h->st, Fl, Fl.K, L, L.K);
    pgxk_rap7(h->st, Fl, Fl.M, L, L.M);
    rc = detect_uniform(h, L);
    if (rc) return rc;
    h->lev.push_back(L);
  }
  // first replicated level: the global grid, and this rank's view of it
  const GridLevel Fl = h->lev.back();
  GridLevel G{};
  G.nx = h->nx >> ld;
  G.ny = D.global_ny >> ld;
  G.n = (G.nx + 1) * (G.ny + 1);
  rc = alloc_level(h, G, true);
  if (rc) return rc;
  const int sxc = G.nx + 1;
  const int Hh = D.global_ny / D.size;
  D.view_own0 = (D.rank * Hh) >> ld;
  D.view_glo = D.rank > 0 ? ghost_mul() : 0;  // the last strip level has 2m ghost rows below and 2m + 1 above
  D.view_H = (Hh >> ld) + (D.rank + 1 == D.size ? 1 : 0);
  D.view_row0 = D.view_own0 - D.view_glo;
  GridLevel& V = D.view;
  V = G;
  V.ny = Fl.ny / 2;
  V.n = sxc * (V.ny + 1);
  if (V.nx * 2 != Fl.nx || D.view_row0 + V.ny > G.ny || V.ny + 1 != D.view_glo + D.view_H + (D.rank + 1 < D.size ? ghost_mul() + 1 : 0)) {
    h->err = "sharded hierarchy: strip view of the first replicated level is inconsistent";
    return PGX_EINVAL;
  }
  const size_t voff = (size_t)D.view_row0 * sxc;
  V.mask = G.mask + voff;
  V.xu = G.xu + voff;
  V.xp = G.xp + voff;
  V.bu = G.bu + voff;
  V.bp = G.bp + voff;
  V.K = V.M = nullptr;  // a view carries vectors and the mask only
  V.Dh = nullptr;
  DALLOC(D.view_S, (size_t)7 * V.n);
  for (int which = 0; which < 2; ++which) {
    pgxk_rap7(h->st, Fl, which ? Fl.M : Fl.K, V, D.view_S);
    double* dst = which ? G.M : G.K;
    pgxk_view_to_global(h->st, 7, V.n, G.n, sxc, D.view_row0, D.view_own0, D.view_H, D.view_S, dst);
    rc = allreduce_dev(h, dst, (size_t)7 * G.n);
    if (rc) return rc;
  }
  {  // Dirichlet mask of the replicated level: injection on the strip, merged like the stencils (setup only)
    uint8_t* vm = nullptr;
    DALLOC(vm, V.n);
    GridLevel Vm = V;
    pgxk_coarse_mask(h->st, Vm, vm, Fl);
    std::vector<uint8_t> hv(V.n);
    HIPCHK(hipMemcpyAsync(hv.data(), vm, V.n, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    std::vector<double> hg(G.n, 0.0);
    for (int J = D.view_own0; J < D.view_own0 + D.view_H; ++J)
      for (int I = 0; I < sxc; ++I) hg[(size_t)J * sxc + I] = hv[(size_t)(J - D.view_row0) * sxc + I];
    HIPCHK(hipMemcpyAsync(G.ru, hg.data(), sizeof(double) * G.n, hipMemcpyHostToDevice, h->st));
    rc = allreduce_dev(h, G.ru, G.n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(hg.data(), G.ru, sizeof(double) * G.n, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    std::vector<uint8_t> gm(G.n);
    for (int v = 0; v < G.n; ++v) gm[v] = hg[v] != 0.0;
    HIPCHK(hipMemcpy(G.mask, gm.data(), G.n, hipMemcpyHostToDevice));
  }
  rc = detect_uniform(h, G);
  if (rc) return rc;
  D.view.interior_free = G.interior_free;  // the view's mask is a row slice of G's
  h->lev.push_back(G);
  int nx = G.nx, ny = G.ny;
  const int min_nx = mg_min_nx();
  while (nx % 2 == 0 && ny % 2 == 0 && nx > min_nx && ny > min_nx) {
    nx /= 2;
    ny /= 2;
    GridLevel L{};
    L.nx = nx;
    L.ny = ny;
    L.n = (nx + 1) * (ny + 1);
    rc = alloc_level(h, L, false);
    if (rc) return rc;
    const GridLevel& Ff = h->lev.back();
    pgxk_coarse_mask(h->st, L, L.mask, Ff);
    pgxk_rap7(h->st, Ff, Ff.K, L, L.K);
    pgxk_rap7(h->st, Ff, Ff.M, L, L.M);
    rc = detect_uniform(h, L);
    if (rc) return rc;
    h->lev.push_back(L);
  }
  setup_tail(h, ld);
  rc = setup_f32(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// pgx_create_curved hands its geometry table to create_impl through this slot (same thread, cleared on return)
static thread_local const double* g_create_geoq = nullptr;

static int create_impl(const pgx_mesh* m, const pgx_problem* p, int device, const pgx_partition* part, pgx_comm* comm,
                       pgx_handle** out) {
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
  if (const char* e = pgx_tune("PGX_XCD_REMAP")) h->xcd_remap = atoi(e) ? 2 : 0;
  if (const char* e = pgx_tune("PGX_TAIL_VERTS")) h->tail_verts = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_PATCH")) h->p2_patch = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_PATCH_NU")) h->patch_nu = std::max(1, atoi(e));
  if (const char* e = pgx_tune("PGX_P2_PATCH_OMEGA")) h->patch_omega = atof(e);
  if (const char* e = pgx_tune("PGX_P2_PATCH_F32")) h->patch_f32 = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_PATCH_SYM")) h->patch_sym = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_RESID_F32")) h->p2_resid_f32 = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_STENCIL")) h->p2st.enable = atoi(e);
  if (const char* e = pgx_tune("PGX_P2_STENCIL_DIST")) h->p2st.dist_enable = atoi(e);
  if (!h->patch_f32) h->patch_sym = 0;
  if (h->patch_f32 == 2 && !h->patch_sym) h->patch_f32 = 1;  // bfloat16 storage exists in the symmetric packing only
  if (const char* e = pgx_tune("PGX_P2_FALLBACK_ITS")) h->p2_fallback_its = std::max(1, atoi(e));
  {
    const char* e = pgx_tune("PGX_TAIL2");  // 0: the round-2 tail kernels (A/B)
    pgxk_mg_tail_select(e ? atoi(e) : 1);
  }
  if (const char* e = pgx_tune("PGX_FUSED_LEGS")) h->fused_legs = atoi(e);
  if (const char* e = pgx_tune("PGX_FUSED_K3")) h->fused_k3 = atoi(e);
  if (const char* e = pgx_tune("PGX_NU_COARSE")) h->nu_coarse = atoi(e);
  if (const char* e = pgx_tune("PGX_SPMV_STREAM")) h->spmv_stream = atoi(e);
  if (const char* e = pgx_tune("PGX_SPMV_BAL")) h->spmv_bal = atoi(e);
  if (const char* e = pgx_tune("PGX_SPMV_DICT")) h->spmv_dict = atoi(e);
  if (const char* e = pgx_tune("PGX_SPMV_STENCIL")) h->spmv_stencil = atoi(e);
  if (const char* e = pgx_tune("PGX_RESID_GRID")) h->resid_grid = atoi(e);
  if (const char* e = pgx_tune("PGX_K6_MAX")) h->k6_max = atoi(e);
  if (const char* e = pgx_tune("PGX_SMOOTH_D32")) h->smooth_d32 = atoi(e);
  if (const char* e = pgx_tune("PGX_HOST_POLL")) h->host_poll = atoi(e);
  if (const char* e = pgx_tune("PGX_LAZY_NORM")) h->lazy_norm = atoi(e);
  if (const char* e = pgx_tune("PGX_Z_F32")) h->z_f32 = atoi(e);
  if (const char* e = pgx_tune("PGX_SPMV_D4")) h->spmv_d4 = atoi(e);
  if (const char* e = pgx_tune("PGX_LEAN_D")) h->lean_d = atoi(e);
  if (const char* e = pgx_tune("PGX_MG_F32")) h->mg_f32 = atoi(e);
  if (const char* e = pgx_tune("PGX_F32_MIN")) h->f32_min = atoi(e);
  if (const char* e = pgx_tune("PGX_F32_RR_MAX")) h->f32_rr_max = atoi(e);
  if (const char* e = pgx_tune("PGX_F32_K6_MAX")) h->f32_k6_max = atoi(e);
  if (const char* e = pgx_tune("PGX_STAG_ITS")) h->stag_its = std::max(2, atoi(e));
  if (const char* e = pgx_tune("PGX_STAG_GAIN")) h->stag_gain = atof(e);
  if (const char* e = pgx_tune("PGX_FUSED_MIN")) h->fused_min = atoi(e);
  if (const char* e = pgx_tune("PGX_CGS_SELECTIVE")) h->cgs_selective = atoi(e);
  if (const char* e = pgx_tune("PGX_CGS_ETA2")) h->cgs_eta2 = atof(e);
  if (const char* e = pgx_tune("PGX_COARSE_SWEEPS")) h->coarse_sweeps = atoi(e);
  auto fail = [&](int rc) {
    g_create_error = h->err;
    pgx_destroy(h);
    return rc;
  };
  h->device = device;
  if (p->degree != 1 && p->degree != 2) {
    h->err = "only Lagrange degree 1 and 2 are implemented";
    return fail(PGX_EINVAL);
  }
  h->degree = p->degree;
  if (p->degree == 2 && (!m->cell_dofs || m->n_dofs <= m->n_vertices)) {
    h->err = "degree 2 needs pgx_mesh.cell_dofs [n_cells][6] and n_dofs = n_vertices + n_edges";
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
  const int nd = h->nd = (p->degree == 2) ? m->n_dofs : n;
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
  if (part) {  // sharded: this mesh is one strip of a global structured mesh (include/pgx.h, "Sharded path")
    pgx_partition pt = *part;
    int32_t row0 = 0, nrows = 0, own0 = 0, nown = 0;
    if (pgx_partition_rows(&pt, &row0, &nrows, &own0, &nown)) {
      h->err = g_create_error;
      return fail(PGX_EINVAL);
    }
    if (!comm || comm->rank != pt.rank || comm->size != pt.size) {
      h->err = "pgx_create_sharded: communicator rank/size differ from the partition's";
      return fail(PGX_EINVAL);
    }
    if (!h->structured || (p->degree != 1 && p->degree != 2)) {
      h->err = "pgx_create_sharded: structured P1 / P2 meshes are sharded (strip decomposition)";
      return fail(PGX_EINVAL);
    }
    if (h->ny + 1 != nrows || h->nx % (1 << pt.dist_levels)) {
      h->err = "pgx_create_sharded: local mesh must hold exactly the vertex rows of pgx_partition_rows, and nx must be "
               "divisible by 2^dist_levels";
      return fail(PGX_EINVAL);
    }
    for (int c = 0; c < nc; ++c) {  // owned cells are addressed as a contiguous range: cells must be ordered by rows
      const int v0 = std::min(m->cells[3 * c], std::min(m->cells[3 * c + 1], m->cells[3 * c + 2]));
      if (v0 / (h->nx + 1) != c / (2 * h->nx)) {
        h->err = "pgx_create_sharded: cells must be ordered row by row (cell 2*(j*nx+i)+t)";
        return fail(PGX_EINVAL);
      }
    }
    Dist& D = h->dist;
    D.on = true;
    D.comm = comm;
    D.rank = pt.rank;
    D.size = pt.size;
    D.ldist = pt.dist_levels;
    D.global_ny = pt.global_ny;
    const int Hh = pt.global_ny / pt.size;
    for (int l = 0; l < D.ldist; ++l) {
      DistLevel d;
      d.g = ghost_mul() << (D.ldist - l);
      d.glo = pt.rank > 0 ? d.g : 0;
      d.H = (Hh >> l) + (pt.rank + 1 == pt.size ? 1 : 0);
      d.ghi = pt.rank + 1 < pt.size ? d.g + 1 : 0;
      D.L.push_back(d);
    }
    const size_t sx = (size_t)h->nx + 1;
    D.own_off = D.L[0].glo * sx;
    D.own_cnt = D.L[0].H * sx;
    const int cell_rows = (pt.rank + 1 == pt.size) ? D.L[0].H - 1 : D.L[0].H;
    D.cell0 = 2 * h->nx * D.L[0].glo;
    D.ncell_own = 2 * h->nx * cell_rows;
    if (p->degree == 2) {
      D.eb = 3 * (size_t)h->nx + 1;
      const size_t last = (size_t)h->ny;  // the last local vertex row carries its nx horizontal edges only
      auto E = [&](size_t row) { return row <= last ? row * D.eb : last * D.eb + (size_t)h->nx; };
      D.eown_off = E(D.L[0].glo);
      D.eown_cnt = E((size_t)D.L[0].glo + D.L[0].H) - D.eown_off;
    }
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
  if (p->degree == 2) {  // P2 tables (basis ordering: SURVEY.md App. A.2)
    QuadTab2& t = h->q2;
    t.nq = p->nq;
    const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
    const int ej[3] = {1, 0, 0}, ek[3] = {2, 2, 1};
    for (int k = 0; k < p->nq; ++k) {
      const double X = p->qpts[2 * k], Y = p->qpts[2 * k + 1];
      const double l[3] = {1.0 - X - Y, X, Y};
      t.w[k] = p->qwts[k];
      for (int i = 0; i < 3; ++i) {
        t.L[k][i] = l[i];
        t.N[k][i] = l[i] * (2.0 * l[i] - 1.0);
        t.N[k][3 + i] = 4.0 * l[ej[i]] * l[ek[i]];
        for (int d = 0; d < 2; ++d) {
          t.dN[k][i][d] = (4.0 * l[i] - 1.0) * dl[i][d];
          t.dN[k][3 + i][d] = 4.0 * (l[ej[i]] * dl[ek[i]][d] + l[ek[i]] * dl[ej[i]][d]);
        }
      }
    }
  }
  // Dirichlet data (dofs of the solution space; the P1 coarse space uses the vertex prefix)
  std::vector<uint8_t> hmask(nd, 0);
  std::vector<double> hg(nd, 0.0);
  for (int k = 0; k < p->n_bc; ++k) {
    const int d = p->bc_dofs[k];
    if (d < 0 || d >= nd) {
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
    DALLOC(h->mask, nd);
    DALLOC(h->gbc, nd);
    DALLOC(h->bphi, nd);
    HIPCHK(hipMemcpy(h->coords, m->coords, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    if (g_create_geoq) {  // isoparametric cells of order 2: weights and inverse Jacobians per quadrature point (pgx_p2.hip)
      if (h->structured || part) {
        h->err = "pgx_create_curved: order-2 geometry is implemented for unstructured, unpartitioned meshes (degree-1 and degree-2 "
                 "fields); structured grids have affine cells by construction";
        return PGX_EINVAL;
      }
      DALLOC(h->geoq, (size_t)5 * nc * p->nq);
      HIPCHK(hipMemcpy(h->geoq, g_create_geoq, sizeof(double) * 5 * (size_t)nc * p->nq, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(h->cells, m->cells, sizeof(int32_t) * 3 * (size_t)nc, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->mask, hmask.data(), nd, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->gbc, hg.data(), sizeof(double) * nd, hipMemcpyHostToDevice));
    if (p->degree == 2) {
      int r2 = build_plan_p2(h, m, hmask);
      if (r2) return r2;
    }
    {  // the P1 plan first: its vertex -> (cell, local vertex) lists also drive the b_phi scatter
      const int r = build_plan(h, m, hmask);
      if (r) return r;
    }
    // b_phi from phi at quadrature points, then phi_q is dropped (it never changes: obstacle_pg.py:107-111)
    double* phi_q = nullptr;
    const size_t nphi = (size_t)nc * p->nq;
    if (hipMalloc((void**)&phi_q, nphi * sizeof(double)) != hipSuccess) {
      h->err = "hipMalloc(phi_q)";
      return PGX_ENOMEM;
    }
    hipError_t e = hipMemcpy(phi_q, p->phi_q, nphi * sizeof(double), hipMemcpyHostToDevice);
    double* stash = nullptr;  // element vectors, summed per dof in list order (no atomics: b_phi is bitwise reproducible)
    if (e == hipSuccess) e = hipMalloc((void**)&stash, sizeof(double) * 8 * (size_t)nc);
    if (e == hipSuccess) {
      if (p->degree == 2)
        pgxk_bphi_p2(h->st, nc, nd, h->cdofs, h->coords, phi_q, h->q2, h->p2_v2c_ptr, h->p2_v2c_ent, stash, h->bphi, h->geoq);
      else
        pgxk_bphi(h->st, nc, n, h->cells, h->coords, phi_q, h->q, h->v2c_ptr, h->v2c_ent, stash, h->bphi, h->geoq);
      e = hipStreamSynchronize(h->st);
    }
    hipFree(phi_q);
    if (stash) hipFree(stash);
    if (e != hipSuccess) {
      h->err = std::string("b_phi assembly: ") + hipGetErrorString(e);
      return PGX_EHIP;
    }
    DALLOC(h->Kv, h->nnz);
    DALLOC(h->Mv, h->nnz);
    DALLOC(h->Dv, h->nnz);
    // (order-2 geometry: per-point geometry for a degree-1 discretisation; the P1 level BELOW a degree-2 one is a preconditioner
    // level and keeps the affine operators of the vertices)
    const double* geo1 = p->degree == 1 ? h->geoq : nullptr;
    pgxk_fill_rows(h->st, 0, n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                   nullptr, h->q, h->Kv, geo1);
    pgxk_fill_rows(h->st, 1, n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                   nullptr, h->q, h->Mv, geo1);
    if (p->degree == 2) {
      DALLOC(h->s_K, h->s_nnz);
      DALLOC(h->s_M, h->s_nnz);
      DALLOC(h->s_D, h->s_nnz);
      pgxk_fill_rows_p2(h->st, 0, nd, h->s_fill_lds, h->s_rowptr, h->p2_v2c_ptr, h->p2_v2c_ent, h->p2_v2c_pos, h->cdofs,
                        h->coords, nullptr, h->q2, h->s_K, h->geoq);
      pgxk_fill_rows_p2(h->st, 1, nd, h->s_fill_lds, h->s_rowptr, h->p2_v2c_ptr, h->p2_v2c_ent, h->p2_v2c_pos, h->cdofs,
                        h->coords, nullptr, h->q2, h->s_M, h->geoq);
      if (h->spmv_dict && h->spmv_bal && h->s_blk) {
        const int rc = build_km_dictionary(h);
        if (rc) return rc;
      }
      DALLOC(h->p2_stash, (size_t)16 * nc);
      DALLOC(h->p2_xu, nd);
      DALLOC(h->p2_xp, nd);
      DALLOC(h->p2_ru, nd);
      DALLOC(h->p2_rp, nd);
      DALLOC(h->c1_bu, n);
      DALLOC(h->c1_bp, n);
      DALLOC(h->c1_xu, n);
      DALLOC(h->c1_xp, n);
    } else {
      h->s_rowptr = h->rowptr;
      h->s_colm = h->colm;
      h->s_K = h->Kv;
      h->s_M = h->Mv;
      h->s_D = h->Dv;
      h->s_nnz = h->nnz;
      h->s_fill_lds = h->fill_lds;
    }
    const size_t n2 = 2 * (size_t)nd;
    DALLOC(h->x, n2);
    DALLOC(h->xk, n2);
    DALLOC(h->F, n2);
    DALLOC(h->dx, n2);
    DALLOC(h->xw, n2);
    DALLOC(h->rhs, n2);
    HIPCHK(hipMemsetAsync(h->x, 0, n2 * sizeof(double), h->st));
    HIPCHK(hipMemsetAsync(h->xk, 0, n2 * sizeof(double), h->st));
    // Krylov basis capacity: 50 vectors cover P1 (restart 30).  The P2 two-level preconditioner leaves outlier
    // modes on the late large-alpha systems whose number grows with N; restarting then stagnates, so P2 keeps
    // up to 300 basis vectors within a 96 GB budget (V and Z) - HBM capacity is what MI355X has plenty of.
    h->restart = 50;
    if (p->degree == 2) {
      // PGX_P2_BASIS_GB (default 24): the sparse LU is the default P2 preconditioner (1-2 iterations per Newton step), the
      // long basis only matters for "pc_type": "pgx_mg"; 96 GB reproduces the pre-LU behaviour
      double budget = 24e9;
      if (const char* e = pgx_tune("PGX_P2_BASIS_GB")) budget = 1e9 * atof(e);
      const double per_vec = 2.0 * (double)n2 * sizeof(double);
      h->restart = (int)std::max(50.0, std::min(300.0, budget / per_vec));
    }
    DALLOC(h->V, (size_t)(h->restart + 1) * n2);
    DALLOC(h->Z, (size_t)h->restart * n2);
    DALLOC(h->w, n2);
    DALLOC(h->d_small, 4 * (h->restart + 2));
    DALLOC(h->partials, (size_t)PGX_RED_BLOCKS * (h->restart + 2));
    DALLOC(h->partials2, ((n2 + PGX_BLOCK - 1) / PGX_BLOCK) * (size_t)62);
    {  // small read-backs (norms, the Hessenberg column, observables): pinned + mapped + coherent, one sequence word at the end
      const size_t nsm = 4 * (size_t)(h->restart + 2);
      HIPCHK(hipHostMalloc((void**)&h->h_small, sizeof(double) * (nsm + 2), hipHostMallocMapped | hipHostMallocCoherent));
      memset(h->h_small, 0, sizeof(double) * (nsm + 2));
      HIPCHK(hipHostGetDevicePointer((void**)&h->h_small_dev, h->h_small, 0));
      h->h_seq = reinterpret_cast<unsigned long long*>(h->h_small + nsm);
      h->h_seq_dev = reinterpret_cast<unsigned long long*>(h->h_small_dev + nsm);
    }
    DALLOC(h->tmp_u, n);
    DALLOC(h->tmp_p, n);
    DALLOC(h->res_u, n);
    DALLOC(h->res_p, n);
    h->obs_blocks = pgxk_observables_blocks(nc);
    DALLOC(h->obs_partials, (size_t)h->obs_blocks * 6);
    DALLOC(h->d_out6, 6);
    if (h->dist.on) {
      DALLOC(h->dist.sb, n2);
      DALLOC(h->dist.wc, 2 * h->dist.nk_field() + 2);
    }
    const int r = h->dist.on ? build_multigrid_dist(h) : build_multigrid(h);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(h->st));
    return PGX_OK;
  };
  rc = up();
  if (rc) return fail(rc);
  *out = h;
  return PGX_OK;
}

extern "C" int pgx_create(const pgx_mesh* m, const pgx_problem* p, int device, pgx_handle** out) {
  return create_impl(m, p, device, nullptr, nullptr, out);
}
extern "C" int pgx_create_curved(const pgx_mesh* m, const pgx_problem* p, const double* geoq, int device, pgx_handle** out) {
  if (!geoq) {
    g_create_error = "pgx_create_curved: null geometry table";
    return PGX_EINVAL;
  }
  g_create_geoq = geoq;
  const int rc = create_impl(m, p, device, nullptr, nullptr, out);
  g_create_geoq = nullptr;
  return rc;
}
extern "C" int pgx_create_lu_dist(const pgx_mesh* m, const pgx_problem* p, pgx_comm* comm, int device, pgx_handle** out) {
  if (!comm) {
    g_create_error = "pgx_create_lu_dist: null communicator";
    return PGX_EINVAL;
  }
  int rc = create_impl(m, p, device, nullptr, nullptr, out);
  if (rc) return rc;
  (*out)->lu_comm = comm;
  if (const char* e = pgx_tune("PGX_CHECK_REPLICAS")) (*out)->check_replicas = atoi(e) != 0;
  return PGX_OK;
}
extern "C" int pgx_create_sharded(const pgx_mesh* m, const pgx_problem* p, const pgx_partition* part, pgx_comm* comm,
                                  int device, pgx_handle** out) {
  if (!part || !comm) {
    g_create_error = "pgx_create_sharded: null partition / communicator";
    return PGX_EINVAL;
  }
  return create_impl(m, p, device, part, comm, out);
}

extern "C" void pgx_destroy(pgx_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->st) hipStreamSynchronize(h->st);
  if (h->cgs_selective && pgx_tune("PGX_CGS_REPORT"))
    fprintf(stderr, "pgx: selective CGS2 skipped the second projection in %ld of %ld Krylov iterations\n", h->cgs_skipped, h->cgs_total);
  if (h->lu) pgx_nd_destroy(h->lu);
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
  *nd = 2 * (int64_t)h->nd;
  return PGX_OK;
}
static int copy_in(pgx_handle* h, double* dst, const double* src) {
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * 2 * (size_t)h->nd, hipMemcpyHostToDevice, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
static int copy_out(pgx_handle* h, double* dst, const double* src) {
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * 2 * (size_t)h->nd, hipMemcpyDeviceToHost, h->st));
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
  pgxk_scale_copy(h->st, 2 * (size_t)h->nd, 1.0, h->x, h->xk);  // a streaming kernel: the runtime's D2D blit ran these 67 MB at 0.5 TB/s
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_zero_state(pgx_handle* h) {
  NEED(h);
  HIPCHK(hipMemsetAsync(h->x, 0, sizeof(double) * 2 * (size_t)h->nd, h->st));
  HIPCHK(hipMemsetAsync(h->xk, 0, sizeof(double) * 2 * (size_t)h->nd, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_owned_range(const pgx_handle* h, int64_t* offset, int64_t* count) {
  if (!h) return PGX_EINVAL;
  if (offset) *offset = h->dist.on ? (int64_t)h->dist.own_off : 0;
  if (count) *count = h->dist.on ? (int64_t)h->dist.own_cnt : (int64_t)(h->degree == 2 ? h->n : h->nd);
  return PGX_OK;
}
extern "C" int pgx_owned_edge_range(const pgx_handle* h, int64_t* offset, int64_t* count) {
  if (!h) return PGX_EINVAL;
  if (offset) *offset = (int64_t)h->n + (h->dist.on ? (int64_t)h->dist.eown_off : 0);
  if (count) *count = h->dist.on ? (int64_t)h->dist.eown_cnt : (int64_t)(h->nd - h->n);
  return PGX_OK;
}
extern "C" int pgx_sync_ghosts(pgx_handle* h) {
  NEED(h);
  if (!h->dist.on) return PGX_OK;
  int rc = halo_solution(h, h->x, h->x + h->nd);
  if (!rc) rc = halo_solution(h, h->xk, h->xk + h->nd);
  if (rc) return rc;
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
// 2-norm of a device vector.  Sharded handles pass OWNED-COMPACT vectors (len = 2 * own_cnt): the squared norms of
// the ranks are summed by one all-reduce before the square root.
// Replicas of a distributed-LU handle (pgx_create_lu_dist) compute every steering scalar redundantly - norms, Gram-Schmidt
// coefficients, observables.  The kernels are deterministic, so the copies agree bitwise today; but a rank that ever took a
// different branch would issue different collectives and RCCL has no timeout.  So the few doubles that DECIDE control flow are
// always taken from rank 0 (one tiny all-reduce of "mine if rank 0 else zero"); only the O(n) residual comparison stays behind
// PGX_CHECK_REPLICAS.  No-op on ordinary and on sharded handles.
static int replica_agree(pgx_handle* h, double* dev, size_t n) {
  if (!h->lu_comm || h->lu_comm->size == 1) return PGX_OK;
  if (h->lu_comm->rank != 0) HIPCHK(hipMemsetAsync(dev, 0, n * sizeof(double), h->st));
  const int rc = h->lu_comm->allreduce(h->st, dev, n);
  if (rc) h->err = "replica agreement: " + h->lu_comm->err;
  return rc;
}

// Small device results -> host WITHOUT a stream synchronisation (round 4; VERDICT r03: "no hipStreamSynchronize between Krylov
// iterations").  A one-workgroup kernel at the end of the enqueued work stores the n doubles into pinned, mapped host memory with
// system-scope stores, fences, and publishes a sequence number; the host polls that word.  What used to cost a D2H copy command, a
// stream synchronisation and the driver's wake-up (15-25 us in which the GPU idles) costs the PCIe write and a cache miss.
__global__ void __launch_bounds__(64) k_publish(int n, const double* __restrict__ src, double* dst, int n2, const double* __restrict__ src2,
                                                double* dst2, unsigned long long seq, unsigned long long* seqp) {
  for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int i = threadIdx.x; i < n2; i += 64) __hip_atomic_store(dst2 + i, src2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();  // one wave: every lane's stores are ordered before lane 0's flag store below
  if (threadIdx.x == 0) __hip_atomic_store(seqp, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// h_small[hoff .. hoff + n) <- dsrc[0 .. n) (and optionally a second range), blocking until the values have arrived
static int fetch_small(pgx_handle* h, const double* dsrc, size_t n, size_t hoff = 0, const double* dsrc2 = nullptr, size_t n2 = 0,
                       size_t hoff2 = 0) {
  if (!h->host_poll) {
    HIPCHK(hipMemcpyAsync(h->h_small + hoff, dsrc, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    if (n2) HIPCHK(hipMemcpyAsync(h->h_small + hoff2, dsrc2, sizeof(double) * n2, hipMemcpyDeviceToHost, h->st));
    HIPCHK(hipStreamSynchronize(h->st));
    return PGX_OK;
  }
  const unsigned long long seq = ++h->seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, h->st, (int)n, dsrc, h->h_small_dev + hoff, (int)n2, dsrc2, h->h_small_dev + hoff2,
                     seq, h->h_seq_dev);
  // Bounded wait: a collective or a kernel ahead of k_publish that never completes leaves hipStreamQuery at NotReady
  // for ever - after PGX_COMM_TIMEOUT seconds (default 120; the transports' own limit) the call fails with PGX_ECOMM / PGX_EHIP and
  // a message instead of spinning.  The poll backs off: a pause per probe, and after ~50 us without an answer a yield per batch of
  // probes, so that eight ranks plus the BLAS threads of an oversubscribed host do not starve each other.
  static const double limit_s = [] {
    const char* t = getenv("PGX_COMM_TIMEOUT");
    return t ? std::max(1.0, atof(t)) : 120.0;
  }();
  std::chrono::steady_clock::time_point t0;
  bool timed = false;
  for (unsigned long spins = 1;; ++spins) {
    if (__atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) == seq) return PGX_OK;
    __builtin_ia32_pause();
    if (spins > 0x2000 && (spins & 0xff) == 0) std::this_thread::yield();
    if ((spins & 0x3fff) == 0) {  // a failed launch or a faulted kernel must not leave the host spinning
      if (!timed) {
        t0 = std::chrono::steady_clock::now();
        timed = true;
      } else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
        h->err = "fetch_small: the device did not publish within PGX_COMM_TIMEOUT (" + std::to_string((int)limit_s) +
                 " s): a kernel or a collective ahead of it on the stream does not complete" + (h->dist.on || h->lu_comm ? " (a peer rank is missing?)" : "");
        return (h->dist.on || h->lu_comm) ? PGX_ECOMM : PGX_EHIP;
      }
      const hipError_t e = hipStreamQuery(h->st);
      if (e == hipSuccess) {
        if (__atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) == seq) return PGX_OK;
        h->err = "fetch_small: the stream drained without publishing (launch failure)";
        return PGX_EHIP;
      }
      if (e != hipErrorNotReady) {
        h->err = std::string("fetch_small: ") + hipGetErrorString(e);
        return PGX_EHIP;
      }
    }
  }
}

static int dev_norm(pgx_handle* h, const double* v, double* out, size_t len = 0) {
  pgxk_multidot(h->st, len ? len : 2 * (size_t)h->nd, 1, v, 0, v, h->partials, h->d_small);
  if (h->dist.on) {
    const int rc = allreduce_dev(h, h->d_small, 1);
    if (rc) return rc;
  }
  {
    const int rc = replica_agree(h, h->d_small, 1);
    if (rc) return rc;
  }
  {
    const int rc = fetch_small(h, h->d_small, 1);
    if (rc) return rc;
  }
  *out = std::sqrt(h->h_small[0]);
  return PGX_OK;
}

static void residual_dev(pgx_handle* h, const double* x, double* F, int with_d = 0) {
  PhaseTimer t(h, 0);
  if (h->degree == 2) {
    pgxk_residual_p2_cells(h->st, h->nc, h->nd, h->cdofs, h->coords, h->mask, h->gbc, x, h->xk, h->alpha, h->f, h->q2,
                           h->p2_v2c_ptr, h->p2_v2c_ent, h->p2_stash, F, h->geoq);
    pgxk_residual_final(h->st, h->nd, h->mask, h->gbc, h->bphi, x, F);
    return;
  }
  // row-parallel, atomic-free, bitwise reproducible; with_d: also fills D(psi) at the same x (Newton driver)
  if (h->resid_grid && h->structured && !h->lev.empty() && h->lev[0].uniform) {  // uniform structured mesh: LDS-staged element blocks
    // with_d == 2 (the Newton loop of a handle whose operator apply is matrix-free and whose preconditioner is the multigrid
    // cycle): the interior rows of D go to the stencil only - their CSR form (56 B per vertex, 235 MB at 2048^2) has no reader
    // until the next pgx_jacobian_fill / pgx_csr_export, which refill it (jac_valid is dropped at the end of the solve)
    pgxk_resid_fill_grid(h->st, with_d ? 1 : 0, h->lev[0], h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells,
                         h->coords, h->mask, h->gbc, h->bphi, x, h->xk, h->alpha, h->f, h->q, F, h->Dv, with_d);
    h->dh_interior = with_d != 0;  // the interior rows of the finest D stencil are in place (consumed by jacobian_dev(have_d))
    if (with_d == 2) h->dv_lean = true;
    if (with_d == 1) h->dv_lean = false;  // a full fill of the CSR rows too
    return;
  }
  pgxk_resid_fill_p1(h->st, with_d, h->n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells,
                     h->coords, h->mask, h->gbc, h->bphi, x, h->xk, h->alpha, h->f, h->q, F, h->Dv, h->geoq);
}

// have_d: D(psi) at this x was already produced by residual_dev(..., with_d=1)
static int jacobian_dev(pgx_handle* h, const double* x, bool have_d = false) {
  if (!(have_d && h->degree == 1)) {
    PhaseTimer t(h, 1);
    if (h->degree == 2) {
      pgxk_fill_rows_p2(h->st, 2, h->nd, h->s_fill_lds, h->s_rowptr, h->p2_v2c_ptr, h->p2_v2c_ent, h->p2_v2c_pos,
                        h->cdofs, h->coords, x + h->nd, h->q2, h->s_D, h->geoq);
      // Galerkin coarse block T^T D_P2 T == D in the P1 basis with the P2 psi (same quadrature), for the P1 hierarchy
      pgxk_fill_rows_p1_Dp2(h->st, h->n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cdofs, h->coords,
                            x + h->nd, h->q2, h->Dv, h->geoq);
    } else if (h->resid_grid && h->structured && !h->lev.empty() && h->lev[0].uniform) {
      // uniform structured mesh: D(psi) comes from the same element kernel the Newton driver uses (its residual output goes to
      // scratch), so pgx_jacobian_fill + pgx_csr_export put THAT kernel under the entry-wise oracle comparison of the parity tests
      pgxk_resid_fill_grid(h->st, 1, h->lev[0], h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                           h->mask, h->gbc, h->bphi, x, h->xk, h->alpha, h->f, h->q, h->w, h->Dv, 1);
      h->dh_interior = true;
      h->dv_lean = false;  // a full fill: CSR rows and stencil
      have_d = true;
    } else {
      pgxk_fill_rows(h->st, 2, h->n, h->fill_lds, h->rowptr, h->v2c_ptr, h->v2c_ent, h->v2c_pos, h->cells, h->coords,
                     x + h->n, h->q, h->Dv, h->geoq);
    }
  }
  if (h->structured) {
    PhaseTimer t(h, 2);
    pgxk_csr_to_stencil_h(h->st, h->n, h->nx + 1, h->rowptr, h->colm, h->Dv, h->lev[0].Dh,
                          (have_d && h->degree == 1 && h->dh_interior) ? h->lev[0].ny : 0);
    h->dh_interior = false;
    if (h->lev[0].Dh32) pgxk_to_float(h->st, (size_t)4 * h->n, h->lev[0].Dh, h->lev[0].Dh32);
    if (h->lev[0].f32) pgxk_f_pack_d(h->st, h->lev[0]);
    if (h->lev[0].Dd4) pgxk_pack_d4(h->st, h->lev[0]);
    const int ld = h->dist.on ? h->dist.ldist : 0;
    for (size_t l = 1; l < h->lev.size(); ++l) {
      if (h->dist.on && (int)l == ld) {  // strip -> replicated level: owned rows + zeros, summed over the ranks
        const Dist& D = h->dist;
        const GridLevel& G = h->lev[l];
#ifdef PGX_DSTEN_FLOAT  // measurement build only (fp32 multigrid D stencils): the merge of the replicated level is fp64
        h->err = "PGX_DSTEN_FLOAT builds do not support sharded handles";
        return PGX_EINVAL;
#else
        pgxk_rap7h(h->st, h->lev[l - 1], h->lev[l - 1].Dh, D.view, D.view_S);
        pgxk_view_to_global(h->st, 4, D.view.n, G.n, G.nx + 1, D.view_row0, D.view_own0, D.view_H, D.view_S, G.Dh);
        const int rc = allreduce_dev(h, G.Dh, (size_t)4 * G.n);
        if (rc) return rc;
#endif
      } else {
        pgxk_rap7h(h->st, h->lev[l - 1], h->lev[l - 1].Dh, h->lev[l], h->lev[l].Dh);
      }
      if (h->lev[l].f32) pgxk_f_pack_d(h->st, h->lev[l]);
    }
  }
  h->jac_valid = true;
  h->patch_fresh = false;
  h->p2st.fresh = false;
  return PGX_OK;
}

// Structured P2 operator apply (pgx_p2st.hip).  First use: the K / M constants of the reference group, checked on every interior
// group; once per Jacobian: the structure-of-arrays copies of D(psi).  Returns false when the CSR kernels have to do the work.
static bool p2st_ready(pgx_handle* h) {
  pgx_handle::P2St& S = h->p2st;
  if (S.state == 0 || !S.select || (h->dist.on && !S.dist_enable) || !h->s_K || !h->s_D) return false;  // strips too (PGX_P2_STENCIL_DIST=0: CSR kernel there)
  static const int nt[4] = {19, 9, 9, 9}, off[4] = {0, 19, 28, 37};
  const size_t G = (size_t)h->n;  // groups = vertices
  if (S.state == 1) {
    S.state = 0;  // until everything below has succeeded
    const std::vector<int32_t>& rp = h->s_h_rowptr;
    double kmax = 0.0, mmax = 0.0;
    for (int t = 0; t < 4; ++t) {
      const int p = rp[S.ref_rows[t]];
      if (hipMemcpy(S.K + off[t], h->s_K + p, sizeof(double) * nt[t], hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(S.tab.M + off[t], h->s_M + p, sizeof(double) * nt[t], hipMemcpyDeviceToHost) != hipSuccess)
        return false;
    }
    for (int e = 0; e < 46; ++e) kmax = std::max(kmax, std::fabs(S.K[e])), mmax = std::max(mmax, std::fabs(S.tab.M[e]));
    S.tab.kmax = kmax, S.tab.mmax = mmax;
    for (int e = 0; e < 46; ++e) S.tab.aK[e] = S.K[e];
    int* d_fail = nullptr;
    if (hipMalloc((void**)&d_fail, sizeof(int)) != hipSuccess) return false;
    hipMemsetAsync(d_fail, 0, sizeof(int), h->st);
    pgxk_p2st_check(h->st, S.tab, h->nx, h->n, S.i0, S.ni, S.j0, S.nj, 1.0, 1e-12, h->s_rowptr, h->s_K, h->s_M, d_fail);
    int fail = 1;
    const hipError_t e1 = hipMemcpyAsync(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, h->st);
    const hipError_t e2 = hipStreamSynchronize(h->st);
    hipFree(d_fail);
    if (e1 != hipSuccess || e2 != hipSuccess || fail) return false;  // not the uniform mesh the table assumes: CSR kernels
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(double) * 46 * G) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    h->allocs.push_back(q);
    S.Dst = (double*)q;
    if (h->p2_resid_f32) {
      if (hipMalloc(&q, sizeof(float) * 46 * G) == hipSuccess) {
        h->allocs.push_back(q);
        S.Dstf = (float*)q;
      } else {
        (void)hipGetLastError();
      }
    }
    S.state = 2;
  }
  if (!S.fresh) {
    pgxk_p2st_pack(h->st, h->nx, h->n, S.i0, S.ni, S.j0, S.nj, G, h->s_rowptr, h->s_D, S.Dst, 0);
    if (S.Dstf) pgxk_p2st_pack(h->st, h->nx, h->n, S.i0, S.ni, S.j0, S.nj, G, h->s_rowptr, h->s_D, S.Dstf, 1);
    S.fresh = true;
  }
  for (int e = 0; e < 46; ++e) S.tab.aK[e] = h->alpha * S.K[e];
  return true;
}
// y = J x (bu == nullptr) or y = b - J x; inner: a residual inside the preconditioner (may read the float copy of D)
static void p2st_apply(pgx_handle* h, const double* xu, const double* xp, const double* bu, const double* bp, double* yu, double* yp,
                       bool inner) {
  pgx_handle::P2St& S = h->p2st;
  const bool f32 = inner && S.Dstf;
  pgxk_p2st_apply(h->st, S.tab, h->nx, h->n, S.i0, S.ni, S.j0, S.nj, (size_t)h->n, f32 ? (const void*)S.Dstf : (const void*)S.Dst, f32, xu,
                  xp, bu, bp, yu, yp);
  pgxk_p2_rows_csr(h->st, S.nframe, S.frame, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_D, h->alpha, h->mask, xu, xp, bu, bp, yu, yp);
}

// y = J x on device vectors of length 2*nd
static void level_apply(pgx_handle* h, int l, int mode, const double* xu, const double* xp, const double* bu,
                        const double* bp, double omega, int first, double* yu, double* yp);
static void spmv_dev(pgx_handle* h, const double* x, double* y) {
  if (h->degree == 2 && h->spmv_stream && h->spmv_bal && p2st_ready(h)) {
    static PgxTuneInt bench_inner("PGX_P2ST_BENCH_INNER", 0);  // measurement hook (tools/p2_spmv_bench.py): time the float-D form
    p2st_apply(h, x, x + h->nd, nullptr, nullptr, y, y + h->nd, bench_inner.get() != 0);
    return;
  }
  if (h->spmv_stencil && h->structured && h->degree == 1) {
    if (h->spmv_stencil == 2)  // A/B: the generic one-thread-per-vertex stencil kernel
      level_apply(h, 0, 0, x, x + h->nd, nullptr, nullptr, 0.0, 0, y, y + h->nd);
    else
      pgxk_st_spmv(h->st, h->lev[0], h->alpha, x, x + h->nd, h->xcd_remap ? 1 : 0, y, y + h->nd);
    return;
  }
  if (h->spmv_stream && h->spmv_bal && h->s_blk)
    pgxk_bspmv_bal(h->st, h->nd, h->s_nblk, h->s_blk, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_code, h->s_tab, h->s_D, h->alpha, h->mask, x,
                   x + h->nd, nullptr, nullptr, h->xcd_remap ? 1 : 0, y, y + h->nd);
  else if (h->spmv_stream && 2 * h->s_fill_lds <= 100 * 1024)
    pgxk_bspmv_stream(h->st, h->nd, h->s_fill_lds, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_D, h->alpha, h->mask, x,
                      x + h->nd, h->xcd_remap ? 1 : 0, y, y + h->nd);
  else
    pgxk_bspmv(h->st, 0, h->nd, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_D, h->alpha, x, x + h->nd, nullptr,
               nullptr, 0.0, h->xcd_remap, y, y + h->nd);
}

static void level_apply(pgx_handle* h, int l, int mode, const double* xu, const double* xp, const double* bu,
                        const double* bp, double omega, int first, double* yu, double* yp) {
  if (l == 0 && !h->structured)  // general mesh: the block-CSR kernel is also the (single-level) smoother
    pgxk_bspmv(h->st, mode, h->n, h->rowptr, h->colm, h->Kv, h->Mv, h->Dv, h->alpha, xu, xp, bu, bp, omega,
               first | h->xcd_remap, yu, yp);
  else
    pgxk_st_apply(h->st, mode, h->lev[l], h->alpha, xu, xp, bu, bp, omega, first | h->xcd_remap, yu, yp);
}

static void vcycle(pgx_handle* h, int l, const double* bu, const double* bp, double* outu, double* outp, int nu, double omega);

// Single-precision legs of the V-cycle on level l (GridLevel::f32; kernels: pgx_mg32.hip).  With (bu, bp) != nullptr the level
// reads its right-hand side from that fp64 pair (its first launch leaves the float2 copy in L.bf) and writes the result to the
// fp64 pair (outu, outp): the finest level, or a level entered from an fp64 one.  Otherwise it finds its right-hand side in L.bf
// (written by the single-precision level above) and returns the buffer that holds its correction.  A level without f32 below gets
// its right-hand side, and hands back its correction, in fp64 (vcycle()).
static const float2* vcycle_f(pgx_handle* h, int l, const double* bu, const double* bp, double* outu, double* outp, int nu,
                              double omega) {
  GridLevel& L = h->lev[l];
  GridLevel& C = h->lev[l + 1];
  // small levels are bound by the latency of a launch, not by its work: all six sweeps of a leg in ONE launch (the pre-smoothing
  // one restricts the residual as well: two launches per level and cycle instead of five)
  const int K = (nu == 6 && L.n <= h->f32_k6_max && L.n <= h->f32_rr_max) ? 6 : ((nu % 3 == 0 && h->fused_k3) ? 3 : 2);
  const int nl = nu / K;
  const int remap = h->xcd_remap ? 1 : 0;
  float2 *cu = L.xf, *ou = L.xf2;
  // levels up to f32_rr_max vertices: the last pre-smoothing launch also restricts the residual of its result (one launch and one
  // pass over D, b, x less per level and cycle; the finest level keeps the separate launch: its halo overhead costs more there)
  const bool fuse = L.n <= h->f32_rr_max;
  float2* const cbf = C.f32 ? C.bf : nullptr;
  double* const cb64u = C.f32 ? nullptr : C.bu;
  double* const cb64p = C.f32 ? nullptr : C.bp;
  {
    const bool rr = fuse && nl == 1;
    const double bscale = (l == 0 && bu) ? h->rhs_scale : 1.0;
    pgxk_f_smooth(h->st, K, 1, L, h->alpha, nullptr, bu, bp, rr ? &C : nullptr, nullptr, nullptr, nullptr, omega, remap, cu, nullptr,
                  nullptr, rr ? cbf : nullptr, rr ? cb64u : nullptr, rr ? cb64p : nullptr, bscale);
  }
  for (int s = 1; s < nl; ++s) {
    const bool rr = fuse && s + 1 == nl;
    pgxk_f_smooth(h->st, K, 0, L, h->alpha, cu, nullptr, nullptr, rr ? &C : nullptr, nullptr, nullptr, nullptr, omega, remap, ou, nullptr,
                  nullptr, rr ? cbf : nullptr, rr ? cb64u : nullptr, rr ? cb64p : nullptr);
    std::swap(cu, ou);
  }
  if (!fuse) pgxk_f_resid_restrict(h->st, L, h->alpha, cu, C, remap, cbf, cb64u, cb64p);
  const float2* cf = nullptr;
  const double *cdu = nullptr, *cdp = nullptr;
  if (C.f32) {
    cf = vcycle_f(h, l + 1, nullptr, nullptr, nullptr, nullptr, nu, omega);
  } else {
    vcycle(h, l + 1, C.bu, C.bp, C.xu, C.xp, nu, omega);
    cdu = C.xu;
    cdp = C.xp;
  }
  for (int s = 0; s < nl; ++s) {
    const bool lastl = s + 1 == nl;
    float2* const zf = (lastl && l == 0) ? h->zf_out : nullptr;  // the cycle's result stays float2, in the caller's buffer
    const bool out64 = lastl && outu && !zf;
    pgxk_f_smooth(h->st, K, 0, L, h->alpha, cu, nullptr, nullptr, s == 0 ? &C : nullptr, s == 0 ? cf : nullptr, s == 0 ? cdu : nullptr,
                  s == 0 ? cdp : nullptr, omega, remap, zf ? zf : ou, out64 ? outu : nullptr, out64 ? outp : nullptr);
    if (zf) return zf;
    std::swap(cu, ou);
  }
  return cu;
}
static inline bool f32_cycle_ok(const pgx_handle* h, int l, int nu) {
  return h->lev[l].f32 && h->fused_legs && l + 1 < (int)h->lev.size() && nu >= 2 && (nu % 2 == 0 || (nu % 3 == 0 && h->fused_k3));
}

// one V(nu,nu) cycle for J_l x = b, zero initial guess, result in (outu,outp)
static void vcycle(pgx_handle* h, int l, const double* bu, const double* bp, double* outu, double* outp, int nu,
                   double omega) {
  GridLevel& L = h->lev[l];
  if (f32_cycle_ok(h, l, nu)) {
    vcycle_f(h, l, bu, bp, outu, outp, nu, omega);
    return;
  }
  if (l > 0 && l == h->tail_start) {  // all remaining levels in ONE launch (k_mg_tail); result in L.xu/L.xp
    h->tail.nu = (h->nu_coarse > 0) ? std::min(nu, h->nu_coarse) : nu;
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
  const int Kf = (nu % 3 == 0 && h->fused_k3) ? 3 : (nu % 2 == 0 ? 2 : 0);  // sweeps per fused launch
  if (h->structured && !last && Kf && h->fused_legs && L.n >= h->fused_min) {
    // per level: nu/K multi-sweep launches | P^T(b - Jx) | nu/K multi-sweep launches (the first one also adds the
    // prolongated coarse correction)   (pgx_kernels.hip, "Fused V-cycle legs", k_st_smoothK)
    GridLevel& C = h->lev[l + 1];
    const int remap = h->xcd_remap ? 1 : 0;
    // small levels are bound by the latency of a launch's dependent phases, not by its work: all six sweeps of a leg in ONE launch
    const int Kl = (Kf == 3 && nu % 6 == 0 && L.n <= h->k6_max && pgxk_st_smooth6_ok(L)) ? 6 : Kf;
    const int nl = nu / Kl;
    double *cu = Bu, *cp = Bp, *ou = Au, *op = Ap;  // current / other buffer pair
    pgxk_st_smoothK(h->st, Kl, 0, L, h->alpha, nullptr, nullptr, nullptr, nullptr, nullptr, bu, bp, omega, remap, cu, cp);
    for (int s = 1; s < nl; ++s) {
      pgxk_st_smoothK(h->st, Kl, 1, L, h->alpha, cu, cp, nullptr, nullptr, nullptr, bu, bp, omega, remap, ou, op);
      std::swap(cu, ou);
      std::swap(cp, op);
    }
    pgxk_st_resid_restrict(h->st, L, h->alpha, cu, cp, bu, bp, C, remap, C.bu, C.bp);
    vcycle(h, l + 1, C.bu, C.bp, C.xu, C.xp, nu, omega);
    pgxk_st_smoothK(h->st, Kl, 1, L, h->alpha, cu, cp, &C, C.xu, C.xp, bu, bp, omega, remap, ou, op);
    std::swap(cu, ou);
    std::swap(cp, op);
    for (int s = 1; s < nl; ++s) {
      pgxk_st_smoothK(h->st, Kl, 1, L, h->alpha, cu, cp, nullptr, nullptr, nullptr, bu, bp, omega, remap, ou, op);
      std::swap(cu, ou);
      std::swap(cp, op);
    }
    if (cu != Au) {
      hipMemcpyAsync(Au, cu, sizeof(double) * L.n, hipMemcpyDeviceToDevice, h->st);
      hipMemcpyAsync(Ap, cp, sizeof(double) * L.n, hipMemcpyDeviceToDevice, h->st);
    }
    return;
  }
  if (l > 0 && h->nu_coarse > 0) nu = std::min(nu, h->nu_coarse);  // small levels are launch-latency bound: fewer sweeps
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

// ------------------------------------------------------------------------------------------------
// Sharded V-cycle (include/pgx.h "Sharded path").  Same sweeps, residuals and transfers as vcycle()
// - the algebra is identical to the single-handle cycle - on strips with ghost rows.  "Validity depth" d of a vector
// means: correct on global rows [a-d, a+H+d] of the strip [a, a+H).  A halo exchange sets d = g (all local rows);
// K Jacobi sweeps in one launch cost K rows (each sweep reads the neighbours' previous values), the fused
// residual+restriction needs d >= 2, x + P x_c needs d >= K for the sweeps that follow.  Exchanges are issued only
// when the depth runs out: with ghost depth 8/4/2 on the three distributed levels and nu = 6 that is 2 + 5 + 7
// exchanges and one all-reduce (coarse right-hand side) per cycle.
// ------------------------------------------------------------------------------------------------
static int vcycle_dist_f(pgx_handle* h, int l, double* bu, double* bp, double* outu, double* outp, int need_out, int nu, double omega,
                         const float2** res);
static int vcycle_dist(pgx_handle* h, int l, double* bu, double* bp, double* outu, double* outp, int need_out, int nu,
                       double omega) {
  Dist& D = h->dist;
  GridLevel& L = h->lev[l];
  if (L.f32 && h->fused_legs) return vcycle_dist_f(h, l, bu, bp, outu, outp, need_out, nu, omega, nullptr);
  const int g = D.L[l].g;
  const int K = (nu % 3 == 0 && g >= 3 && h->fused_k3) ? 3 : 2;
  if (nu % K) {
    h->err = "sharded V-cycle: mg_nu must be even (or a multiple of 3 with deep enough ghost rows)";
    return PGX_EINVAL;
  }
  const int remap = h->xcd_remap ? 1 : 0;
  const int nl = nu / K;
  int rc = halo_level(h, l, bu, bp);  // restriction / the Krylov vector are correct on owned rows only
  if (rc) return rc;
  double *cu = (l == 0) ? h->tmp_u : L.xu2, *cp = (l == 0) ? h->tmp_p : L.xp2, *ou = outu, *op = outp;
  int xv;  // validity depth of (cu, cp)
  pgxk_st_smoothK(h->st, K, 0, L, h->alpha, nullptr, nullptr, nullptr, nullptr, nullptr, bu, bp, omega, remap, cu, cp);
  xv = g - K;  // first sweep from zero is pointwise: valid where b and the operator are (depth g-1)
  auto more = [&](const GridLevel* C, const double* ccu, const double* ccp) -> int {
    if (xv < K) {
      const int r = halo_level(h, l, cu, cp);
      if (r) return r;
      xv = g;
    }
    pgxk_st_smoothK(h->st, K, 1, L, h->alpha, cu, cp, C, ccu, ccp, bu, bp, omega, remap, ou, op);
    std::swap(cu, ou);
    std::swap(cp, op);
    xv -= K;
    return PGX_OK;
  };
  for (int s = 1; s < nl; ++s)
    if ((rc = more(nullptr, nullptr, nullptr))) return rc;
  if (xv < 2) {  // P^T (b - J x) on the owned coarse rows reads the residual one row out, i.e. x two rows out
    if ((rc = halo_level(h, l, cu, cp))) return rc;
    xv = g;
  }
  const GridLevel* C;
  if (l + 1 < D.ldist) {
    GridLevel& Cl = h->lev[l + 1];
    pgxk_st_resid_restrict(h->st, L, h->alpha, cu, cp, bu, bp, Cl, remap, Cl.bu, Cl.bp);
    if ((rc = vcycle_dist(h, l + 1, Cl.bu, Cl.bp, Cl.xu, Cl.xp, D.L[l + 1].g, nu, omega))) return rc;
    C = &Cl;
  } else {
    // onto the replicated level: every rank restricts into its view, clears the view's ghost rows, and the all-reduce
    // assembles the global right-hand side (exactly one non-zero contribution per entry)
    GridLevel& G = h->lev[l + 1];
    const GridLevel& V = D.view;
    const size_t sxc = (size_t)G.nx + 1;
    if (hipMemsetAsync(G.bu, 0, sizeof(double) * 2 * (size_t)G.n, h->st) != hipSuccess) return PGX_EHIP;
    pgxk_st_resid_restrict(h->st, L, h->alpha, cu, cp, bu, bp, V, remap, V.bu, V.bp);
    const size_t lo = (size_t)D.view_glo * sxc, hi0 = (size_t)(D.view_glo + D.view_H) * sxc,
                 hi = (size_t)V.n - hi0;
    double* const halves[2] = {V.bu, V.bp};
    for (double* b : halves) {
      if (lo) hipMemsetAsync(b, 0, sizeof(double) * lo, h->st);
      if (hi) hipMemsetAsync(b + hi0, 0, sizeof(double) * hi, h->st);
    }
    if ((rc = allreduce_dev(h, G.bu, 2 * (size_t)G.n))) return rc;
    vcycle(h, l + 1, G.bu, G.bp, G.xu, G.xp, nu, omega);  // identical work on every rank
    C = &V;
  }
  // x + P x_c is correct to depth min(xv, g): the coarse correction covers every local row
  if ((rc = more(C, C->xu, C->xp))) return rc;
  for (int s = 1; s < nl; ++s)
    if ((rc = more(nullptr, nullptr, nullptr))) return rc;
  if (xv < need_out) {
    if ((rc = halo_level(h, l, cu, cp))) return rc;
    xv = g;
  }
  if (cu != outu) {
    hipMemcpyAsync(outu, cu, sizeof(double) * L.n, hipMemcpyDeviceToDevice, h->st);
    hipMemcpyAsync(outp, cp, sizeof(double) * L.n, hipMemcpyDeviceToDevice, h->st);
  }
  return PGX_OK;
}

// The sharded cycle on a single-precision strip level (GridLevel::f32): vcycle_dist with the kernels of pgx_mg32.hip.  Same depth
// bookkeeping; a halo exchange moves ONE float2 field (8 bytes per vertex) instead of two fp64 ones.  (bu, bp) != nullptr: fp64
// right-hand side in (the Krylov vector), fp64 result out; else L.bf in, *res = the buffer that holds the correction.
static int vcycle_dist_f(pgx_handle* h, int l, double* bu, double* bp, double* outu, double* outp, int need_out, int nu, double omega,
                         const float2** res) {
  Dist& D = h->dist;
  GridLevel& L = h->lev[l];
  const int g = D.L[l].g;
  const int K = (nu % 3 == 0 && g >= 3 && h->fused_k3) ? 3 : 2;
  if (nu % K) {
    h->err = "sharded V-cycle: mg_nu must be even (or a multiple of 3 with deep enough ghost rows)";
    return PGX_EINVAL;
  }
  const int remap = h->xcd_remap ? 1 : 0;
  const int nl = nu / K;
  int rc = bu ? halo_level(h, l, bu, bp) : halo_level_f(h, l, L.bf);  // the right-hand side is correct on owned rows only
  if (rc) return rc;
  float2 *cu = L.xf, *ou = L.xf2;
  pgxk_f_smooth(h->st, K, 1, L, h->alpha, nullptr, bu, bp, nullptr, nullptr, nullptr, nullptr, omega, remap, cu, nullptr, nullptr, nullptr,
                nullptr, nullptr, (l == 0 && bu) ? h->rhs_scale : 1.0);
  int xv = g - K;  // validity depth of cu
  bool in64 = false;  // the current iterate already sits in (outu, outp)
  auto more = [&](const GridLevel* C, const float2* cf, const double* cdu, const double* cdp, bool last) -> int {
    if (xv < K) {
      const int r = halo_level_f(h, l, cu);
      if (r) return r;
      xv = g;
    }
    const bool out64 = last && outu;
    pgxk_f_smooth(h->st, K, 0, L, h->alpha, cu, nullptr, nullptr, C, cf, cdu, cdp, omega, remap, ou, out64 ? outu : nullptr,
                  out64 ? outp : nullptr);
    std::swap(cu, ou);
    in64 = out64;
    xv -= K;
    return PGX_OK;
  };
  for (int s = 1; s < nl; ++s)
    if ((rc = more(nullptr, nullptr, nullptr, nullptr, false))) return rc;
  if (xv < 2) {  // P^T (b - J x) on the owned coarse rows reads the residual one row out, i.e. x two rows out
    if ((rc = halo_level_f(h, l, cu))) return rc;
    xv = g;
  }
  const GridLevel* C;
  const float2* cf = nullptr;
  const double *cdu = nullptr, *cdp = nullptr;
  if (l + 1 < D.ldist) {
    GridLevel& Cl = h->lev[l + 1];
    if (Cl.f32) {
      pgxk_f_resid_restrict(h->st, L, h->alpha, cu, Cl, remap, Cl.bf, nullptr, nullptr);
      if ((rc = vcycle_dist_f(h, l + 1, nullptr, nullptr, nullptr, nullptr, D.L[l + 1].g, nu, omega, &cf))) return rc;
    } else {
      pgxk_f_resid_restrict(h->st, L, h->alpha, cu, Cl, remap, nullptr, Cl.bu, Cl.bp);
      if ((rc = vcycle_dist(h, l + 1, Cl.bu, Cl.bp, Cl.xu, Cl.xp, D.L[l + 1].g, nu, omega))) return rc;
      cdu = Cl.xu;
      cdp = Cl.xp;
    }
    C = &Cl;
  } else {
    // onto the replicated level (fp64 right-hand side: it is summed over the ranks): every rank restricts into its view, clears the
    // view's ghost rows, and the all-reduce assembles the global right-hand side (exactly one non-zero contribution per entry)
    GridLevel& G = h->lev[l + 1];
    const GridLevel& V = D.view;
    const size_t sxc = (size_t)G.nx + 1;
    if (hipMemsetAsync(G.bu, 0, sizeof(double) * 2 * (size_t)G.n, h->st) != hipSuccess) return PGX_EHIP;
    pgxk_f_resid_restrict(h->st, L, h->alpha, cu, V, remap, nullptr, V.bu, V.bp);
    const size_t lo = (size_t)D.view_glo * sxc, hi0 = (size_t)(D.view_glo + D.view_H) * sxc, hi = (size_t)V.n - hi0;
    double* const halves[2] = {V.bu, V.bp};
    for (double* b : halves) {
      if (lo) hipMemsetAsync(b, 0, sizeof(double) * lo, h->st);
      if (hi) hipMemsetAsync(b + hi0, 0, sizeof(double) * hi, h->st);
    }
    if ((rc = allreduce_dev(h, G.bu, 2 * (size_t)G.n))) return rc;
    vcycle(h, l + 1, G.bu, G.bp, G.xu, G.xp, nu, omega);  // identical work on every rank
    C = &V;
    cdu = V.xu;
    cdp = V.xp;
  }
  // x + P x_c is correct to depth min(xv, g): the coarse correction covers every local row
  if ((rc = more(C, cf, cdu, cdp, nl == 1))) return rc;
  for (int s = 1; s < nl; ++s)
    if ((rc = more(nullptr, nullptr, nullptr, nullptr, s + 1 == nl))) return rc;
  if (xv < need_out) {
    if ((rc = in64 ? halo_level(h, l, outu, outp) : halo_level_f(h, l, cu))) return rc;
    xv = g;
  }
  if (res) *res = cu;
  return PGX_OK;
}

// owned entries of a local (u | psi) vector <-> owned-compact vector [u_owned | psi_owned]
static void gather_owned(pgx_handle* h, const double* loc, double* cmp) {
  const Dist& D = h->dist;
  const size_t nf = D.nk_field(), nd = (size_t)h->nd, nv = (size_t)h->n;
  for (int f = 0; f < 2; ++f) {
    hipMemcpyAsync(cmp + f * nf, loc + f * nd + D.own_off, sizeof(double) * D.own_cnt, hipMemcpyDeviceToDevice, h->st);
    if (D.eown_cnt)
      hipMemcpyAsync(cmp + f * nf + D.own_cnt, loc + f * nd + nv + D.eown_off, sizeof(double) * D.eown_cnt, hipMemcpyDeviceToDevice,
                     h->st);
  }
}
static void scatter_owned(pgx_handle* h, const double* cmp, double* loc) {
  const Dist& D = h->dist;
  const size_t nf = D.nk_field(), nd = (size_t)h->nd, nv = (size_t)h->n;
  for (int f = 0; f < 2; ++f) {
    hipMemcpyAsync(loc + f * nd + D.own_off, cmp + f * nf, sizeof(double) * D.own_cnt, hipMemcpyDeviceToDevice, h->st);
    if (D.eown_cnt)
      hipMemcpyAsync(loc + f * nd + nv + D.eown_off, cmp + f * nf + D.own_cnt, sizeof(double) * D.eown_cnt, hipMemcpyDeviceToDevice,
                     h->st);
  }
}

// P2: two-level cycle.  Smoother = collective damped Jacobi on the P2 block CSR (k_bspmv<2>); coarse space =
// the P1 subspace with its full multigrid hierarchy (one V-cycle); T = P1->P2 interpolation.
// Patch data of the P2 level: tables on first use, inverses once per Jacobian (alpha and D(psi) change every Newton step)
static int ensure_patches(pgx_handle* h) {
  const int NN = h->patch_nn, nv = h->n, nd = h->nd;
  if (!h->pdof) {
    DALLOC(h->pdof, (size_t)nv * NN);
    DALLOC(h->ppos, (size_t)nv * NN * NN);
    {
      uint8_t* q = nullptr;
      DALLOC(q, pgxk_patch_inverse_bytes(nv, NN, h->patch_f32, h->patch_sym));
      h->pinv = q;
    }
    if (h->p2_resid_f32) DALLOC(h->s_Df, (size_t)h->s_nnz);
    DALLOC(h->p2_su, (size_t)2 * (nd - nv));
    DALLOC(h->p2_sp, (size_t)2 * (nd - nv));
    HIPCHK(hipMemcpy(h->pdof, h->patch_dof_host.data(), sizeof(int32_t) * h->patch_dof_host.size(), hipMemcpyHostToDevice));
    std::vector<int32_t>().swap(h->patch_dof_host);
    pgxk_patch_positions(h->st, nv, NN, h->pdof, h->s_rowptr, h->s_colm, h->ppos);
  }
  if (!h->patch_fresh) {
    pgxk_patch_invert(h->st, nv, NN, h->pdof, h->ppos, h->s_K, h->s_M, h->s_D, h->mask, h->alpha, h->pinv, h->patch_f32, h->patch_sym);
    if (h->s_Df) pgxk_to_float(h->st, (size_t)h->s_nnz, h->s_D, h->s_Df);
    h->patch_fresh = true;
  }
  return PGX_OK;
}

// Two-level cycle with the vertex-star patch smoother (round 3): patch_nu additive sweeps | P1 hierarchy on T^T (b - J x) |
// patch_nu sweeps.  A sweep = residual (block-CSR SpMV) + one pass over the patch inverses.
static void pcycle_p2_patch(pgx_handle* h, const double* bu, const double* bp, double* xu, double* xp, int nu, double omega) {
  const int nd = h->nd, nv = h->n, NN = h->patch_nn;
  const bool st_apply = h->spmv_bal && p2st_ready(h);
  auto resid = [&]() {
    if (st_apply)
      p2st_apply(h, xu, xp, bu, bp, h->p2_ru, h->p2_rp, true);
    else if (h->spmv_bal && h->s_blk)
      pgxk_bspmv_bal(h->st, nd, h->s_nblk, h->s_blk, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_code, h->s_tab, h->s_D, h->alpha, h->mask, xu, xp, bu,
                     bp, h->xcd_remap ? 1 : 0, h->p2_ru, h->p2_rp, h->s_Df);
    else
      pgxk_bspmv(h->st, 1, nd, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_D, h->alpha, xu, xp, bu, bp, 0.0, h->xcd_remap,
                 h->p2_ru, h->p2_rp);
  };
  auto patch = [&](const double* ru, const double* rp) {
    pgxk_patch_sweep(h->st, nv, NN, nv, nd, h->pdof, h->edge_ends, h->pinv, h->patch_f32, h->patch_sym, ru, rp, h->patch_omega, xu, xp, h->p2_su, h->p2_sp);
  };
  hipMemsetAsync(xu, 0, sizeof(double) * nd, h->st);
  hipMemsetAsync(xp, 0, sizeof(double) * nd, h->st);
  patch(bu, bp);  // x = 0: the residual is b
  for (int s2 = 1; s2 < h->patch_nu; ++s2) {
    resid();
    patch(h->p2_ru, h->p2_rp);
  }
  resid();
  pgxk_p2_restrict(h->st, nv, nd, h->v2e_ptr, h->v2e, h->mask, h->p2_ru, h->p2_rp, h->c1_bu, h->c1_bp);
  vcycle(h, 0, h->c1_bu, h->c1_bp, h->c1_xu, h->c1_xp, nu, omega);
  pgxk_p2_prolong_add(h->st, nv, nd, h->edge_ends, h->c1_xu, h->c1_xp, xu, xp);
  for (int s2 = 0; s2 < h->patch_nu; ++s2) {
    resid();
    patch(h->p2_ru, h->p2_rp);
  }
}

// The same cycle on a strip (sharded P2, round 3).  Validity depth = number of ghost vertex rows on which a local vector equals the
// global one.  An exchange (halo_solution) restores depth g; a residual b - J x reads one row further out than it writes (-1); a patch
// solve needs the residual on its whole star and an edge average both end patches (-1): a sweep = residual + patches costs 2 rows.
// The P1 hierarchy below is vcycle_dist, which exchanges for itself.  Exchanges are issued only when the depth runs out - with the
// default ghost depth (24 rows at three distributed levels) that is the one exchange of the right-hand side.
static int pcycle_p2_patch_dist(pgx_handle* h, double* bu, double* bp, double* xu, double* xp, int nu, double omega) {
  const int nd = h->nd, nv = h->n, NN = h->patch_nn;
  const int g = h->dist.L[0].g;
  int rc = halo_solution(h, bu, bp);  // the Krylov vector is correct on owned dofs only
  if (rc) return rc;
  int xv = 0;  // validity depth of (xu, xp)
  auto need = [&](int depth) -> int {
    if (xv >= depth) return PGX_OK;
    const int r = halo_solution(h, xu, xp);
    xv = g;
    return r;
  };
  const bool st_apply = h->spmv_bal && p2st_ready(h);
  auto resid = [&]() {
    if (st_apply)
      p2st_apply(h, xu, xp, bu, bp, h->p2_ru, h->p2_rp, true);
    else
      pgxk_bspmv_bal(h->st, nd, h->s_nblk, h->s_blk, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_code, h->s_tab, h->s_D, h->alpha, h->mask, xu, xp, bu,
                     bp, h->xcd_remap ? 1 : 0, h->p2_ru, h->p2_rp, h->s_Df);
    xv -= 1;
  };
  auto patch = [&](const double* ru, const double* rp) {
    pgxk_patch_sweep(h->st, nv, NN, nv, nd, h->pdof, h->edge_ends, h->pinv, h->patch_f32, h->patch_sym, ru, rp, h->patch_omega, xu, xp, h->p2_su, h->p2_sp);
    xv -= 1;
  };
  if (!h->s_blk) {
    h->err = "sharded P2: no balanced SpMV blocks";
    return PGX_ESTATE;
  }
  hipMemsetAsync(xu, 0, sizeof(double) * nd, h->st);
  hipMemsetAsync(xp, 0, sizeof(double) * nd, h->st);
  xv = g;  // zero is right everywhere
  patch(bu, bp);
  for (int s2 = 1; s2 < h->patch_nu; ++s2) {
    if ((rc = need(3))) return rc;
    resid();
    patch(h->p2_ru, h->p2_rp);
  }
  if ((rc = need(3))) return rc;
  resid();  // valid to depth >= 2: the P1 restriction reads the residual one row out on the owned coarse rows
  pgxk_p2_restrict(h->st, nv, nd, h->v2e_ptr, h->v2e, h->mask, h->p2_ru, h->p2_rp, h->c1_bu, h->c1_bp);
  if ((rc = vcycle_dist(h, 0, h->c1_bu, h->c1_bp, h->c1_xu, h->c1_xp, g, nu, omega))) return rc;
  pgxk_p2_prolong_add(h->st, nv, nd, h->edge_ends, h->c1_xu, h->c1_xp, xu, xp);
  xv = std::min(xv, g - 1);  // the edge part of T x_c reads both end vertices
  for (int s2 = 0; s2 < h->patch_nu; ++s2) {
    if ((rc = need(3))) return rc;
    resid();
    patch(h->p2_ru, h->p2_rp);
  }
  return need(2);  // the operator apply that follows reads one row beyond the strip
}

static void pcycle_p2(pgx_handle* h, const double* bu, const double* bp, double* outu, double* outp, int nu,
                      double omega) {
  const int nd = h->nd;
  const int nu2 = nu + 1;
  const double om2 = 0.75 * omega;
  auto app = [&](int mode, const double* xu, const double* xp, int first, double* yu, double* yp) {
    pgxk_bspmv(h->st, mode, nd, h->s_rowptr, h->s_colm, h->s_K, h->s_M, h->s_D, h->alpha, xu, xp, bu, bp, om2,
               first | h->xcd_remap, yu, yp);
  };
  double *Au = outu, *Ap = outp, *Bu = h->p2_xu, *Bp = h->p2_xp;
  bool toA = false;  // 2*nu2 sweeps in total (even): start in B so that the last one lands in A
  const double *cu = nullptr, *cp = nullptr;
  auto sweep = [&](int first) {
    double* tu = toA ? Au : Bu;
    double* tp = toA ? Ap : Bp;
    app(2, cu, cp, first, tu, tp);
    cu = tu;
    cp = tp;
    toA = !toA;
  };
  for (int s2 = 0; s2 < nu2; ++s2) sweep(s2 == 0);
  app(1, cu, cp, 0, h->p2_ru, h->p2_rp);
  pgxk_p2_restrict(h->st, h->n, nd, h->v2e_ptr, h->v2e, h->mask, h->p2_ru, h->p2_rp, h->c1_bu, h->c1_bp);
  vcycle(h, 0, h->c1_bu, h->c1_bp, h->c1_xu, h->c1_xp, nu, omega);
  pgxk_p2_prolong_add(h->st, h->n, nd, h->edge_ends, h->c1_xu, h->c1_xp, (double*)cu, (double*)cp);
  for (int s2 = 0; s2 < nu2; ++s2) sweep(0);
  if (cu != Au) {  // odd nu: final sweep landed in the scratch pair
    hipMemcpyAsync(Au, cu, sizeof(double) * nd, hipMemcpyDeviceToDevice, h->st);
    hipMemcpyAsync(Ap, cp, sizeof(double) * nd, hipMemcpyDeviceToDevice, h->st);
  }
}

// Values of the mixed Newton matrix [[alpha K, M],[M, -D]] with the Dirichlet rows/columns of the u block replaced by
// identity (the contract of problem.py:69-77), in the CSR layout: row i -> [cols_s(i) | nd + cols_s(i)], row nd+i likewise.
__global__ void k_mixed_vals(int nd, int nnz, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colm,
                             const double* __restrict__ K, const double* __restrict__ M, const double* __restrict__ D,
                             double alpha, const uint8_t* __restrict__ mask, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nd) return;
  const int b = rowptr[i], e = rowptr[i + 1], len = e - b;
  const bool rowbc = mask[i] != 0;
  double* uu = out + 2 * (size_t)b;
  double* up = uu + len;
  double* pu = out + 2 * (size_t)nnz + 2 * (size_t)b;
  double* pp = pu + len;
  for (int k = b; k < e; ++k) {
    const int32_t cm = colm[k];
    const bool colbc = cm < 0;
    const int c = cm & 0x7fffffff;
    const double m = M[k];
    uu[k - b] = (rowbc || colbc) ? ((rowbc && c == i) ? 1.0 : 0.0) : alpha * K[k];
    up[k - b] = rowbc ? 0.0 : m;
    pu[k - b] = colbc ? 0.0 : m;
    pp[k - b] = -D[k];
  }
}

// symbolic phase of the direct solver, once per handle (first Newton solve that asks for pc_type lu)
static int ensure_lu(pgx_handle* h) {
  if (h->lu) return PGX_OK;
  if (h->dist.on) {
    h->err = "pc_type lu is not available on a sharded handle";
    return PGX_EINVAL;
  }
  const int nd = h->nd, n = h->n;
  const std::vector<int32_t>& hr = (h->degree == 2) ? h->s_h_rowptr : h->h_rowptr;
  const std::vector<int32_t>& hc = (h->degree == 2) ? h->s_h_col : h->h_col;
  const int64_t nnz = hr[nd];
  if ((int64_t)4 * nnz > 0x7fffffff) {
    h->err = "pc_type lu: mixed matrix exceeds int32 nnz";
    return PGX_EINVAL;
  }
  std::vector<int32_t> rp(2 * (size_t)nd + 1), cl(4 * (size_t)nnz), nod(2 * (size_t)nd);
  for (int i = 0; i < nd; ++i) {
    const int b = hr[i], len = hr[i + 1] - b;
    rp[i] = 2 * b;
    rp[nd + i] = (int32_t)(2 * nnz + 2 * b);
    for (int k = 0; k < len; ++k) {
      const int32_t c = hc[b + k] & 0x7fffffff;
      if (k > 0 && (hc[b + k - 1] & 0x7fffffff) >= c) {
        h->err = "pc_type lu: scalar pattern is not sorted";
        return PGX_EINVAL;
      }
      cl[2 * (size_t)b + k] = c;
      cl[2 * (size_t)b + len + k] = nd + c;
      cl[2 * (size_t)nnz + 2 * (size_t)b + k] = c;
      cl[2 * (size_t)nnz + 2 * (size_t)b + len + k] = nd + c;
    }
    nod[i] = nod[nd + i] = i;
  }
  rp[2 * (size_t)nd] = (int32_t)(4 * nnz);
  std::vector<double> xy(2 * (size_t)nd);
  HIPCHK(hipMemcpy(xy.data(), h->coords, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
  if (h->degree == 2) {
    std::vector<int32_t> ends(2 * (size_t)(nd - n));
    HIPCHK(hipMemcpy(ends.data(), h->edge_ends, sizeof(int32_t) * ends.size(), hipMemcpyDeviceToHost));
    for (int e = 0; e < nd - n; ++e)
      for (int d = 0; d < 2; ++d) xy[2 * (size_t)(n + e) + d] = 0.5 * (xy[2 * (size_t)ends[2 * e] + d] + xy[2 * (size_t)ends[2 * e + 1] + d]);
  }
  pgx_nd_matrix A{};
  A.n = 2 * (int64_t)nd;
  A.rowptr = rp.data();
  A.col = cl.data();
  A.n_nodes = nd;
  A.node_of_dof = nod.data();
  A.dim = 2;
  A.node_coords = xy.data();
  A.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) A.leaf_nodes = atoi(e);
  int rc = h->lu_comm ? pgx_nd_create_dist(&A, h->lu_comm, h->device, (void*)h->st, &h->lu)
                      : pgx_nd_create(&A, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("pc_type lu: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  // [[alpha K, M], [M, -D(psi)]] with identity Dirichlet rows AND columns is symmetric: L D L^T in LU clothing, half the flops
  // (include/pgx_nd.h; ignored on a distributed handle); the factorisation preconditions FGMRES on the exact operator either way
  pgx_nd_set_symmetric(h->lu, 1);
  DALLOC(h->Jmix, 4 * (size_t)nnz);
  return PGX_OK;
}

// replicas of pgx_create_lu_dist: overwrite dev[0:n) with rank 0's copy (all-reduce of "mine if rank 0 else zero")
static int replica_bcast(pgx_handle* h, double* dev, size_t n) {
  if (!h->lu_comm || h->lu_comm->size == 1) return PGX_OK;
  if (h->lu_comm->rank != 0) HIPCHK(hipMemsetAsync(dev, 0, n * sizeof(double), h->st));
  const int rc = h->lu_comm->allreduce(h->st, dev, n);
  if (rc) h->err = "replica broadcast: " + h->lu_comm->err;
  return rc;
}
// The replicas assemble redundantly and every assembly kernel is atomic-free (fixed summation order), so their residuals are
// bitwise identical and nothing needs to be broadcast.  PGX_CHECK_REPLICAS=1 turns that claim into a run-time assertion
// (tests): rank 0's copy travels to every rank and must equal the local one exactly.  Collective; h->w is the scratch.
static int replica_check(pgx_handle* h, const double* dev, size_t n, const char* what) {
  if (!h->lu_comm || h->lu_comm->size == 1 || !h->check_replicas) return PGX_OK;
  HIPCHK(hipMemcpyAsync(h->w, dev, n * sizeof(double), hipMemcpyDeviceToDevice, h->st));
  int rc = replica_bcast(h, h->w, n);
  if (rc) return rc;
  pgxk_axpy(h->st, n, -1.0, dev, h->w);
  pgxk_multidot(h->st, n, 1, h->w, 0, h->w, h->partials, h->d_small);
  if ((rc = h->lu_comm->allreduce(h->st, h->d_small, 1))) {  // sum of the ranks' squared differences: one verdict for all
    h->err = "replica check: " + h->lu_comm->err;
    return rc;
  }
  HIPCHK(hipMemcpyAsync(h->h_small, h->d_small, sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(hipStreamSynchronize(h->st));
  if (h->h_small[0] != 0.0) {
    h->err = std::string("replicas of a distributed-LU handle disagree on ") + what;
    return PGX_ECOMM;
  }
  return PGX_OK;
}

static int lu_factor(pgx_handle* h) {
  PhaseTimer t(h, 2);
  hipLaunchKernelGGL(k_mixed_vals, dim3((h->nd + 127) / 128), dim3(128), 0, h->st, h->nd, h->s_nnz, h->s_rowptr, h->s_colm,
                     h->s_K, h->s_M, h->s_D, h->alpha, h->mask, h->Jmix);
  int rc = pgx_nd_factor(h->lu, h->Jmix, 1);
  if (rc) h->err = std::string("pc_type lu: ") + pgx_nd_last_error(h->lu);
  return rc;
}

// FGMRES keeps its Z_j as float2 fields (round 5) where they leave the single-precision cycle and the operator apply is the
// matrix-free stencil kernel: one predicate for the solver and for pgx_spmv_bench, which replays the solver's sequence
static inline bool z_f32_active(const pgx_handle* h, int nu) {
  return h->z_f32 && !h->dist.on && !h->lu_active && h->degree == 1 && h->structured && !h->lev.empty() && h->lev[0].uniform &&
         h->spmv_stencil == 1 && f32_cycle_ok(h, 0, nu);
}

static int precond(pgx_handle* h, const double* b, double* z, int nu, double omega) {
  if (h->lu_active) {
    int rc = pgx_nd_solve(h->lu, b, z, 1);
    if (rc) h->err = std::string("pc_type lu: ") + pgx_nd_last_error(h->lu);
    return rc;
  }
  if (h->dist.on) {  // b is owned-compact, z local (owned + ghost rows, correct at least one row beyond the strip)
    scatter_owned(h, b, h->dist.sb);
    ++h->dist.n_vcycle;
    if (h->degree == 2) {
      if (!h->patch_nn) {
        h->err = "sharded P2 needs the vertex-star patch smoother (vertex degree <= 7)";
        return PGX_EINVAL;
      }
      int rc = ensure_patches(h);
      if (rc) return rc;
      return pcycle_p2_patch_dist(h, h->dist.sb, h->dist.sb + h->nd, z, z + h->nd, nu, omega);
    }
    return vcycle_dist(h, 0, h->dist.sb, h->dist.sb + h->n, z, z + h->n, 1, nu, omega);
  }
  if (h->degree == 2 && h->p2_patch && h->patch_nn) {  // vertex-star patch smoother on the P2 level, P1 hierarchy below
    const int rc = ensure_patches(h);
    if (rc) return rc;
    pcycle_p2_patch(h, b, b + h->nd, z, z + h->nd, nu, omega);
  } else if (h->degree == 2)  // round-1 cycle: point-collective Jacobi on the P2 level (meshes with vertex degree > 7, A/B)
    pcycle_p2(h, b, b + h->nd, z, z + h->nd, nu, omega);
  else
    vcycle(h, 0, b, b + h->n, z, z + h->n, nu, omega);
  return PGX_OK;
}

// FGMRES(restart) on J dx = b, right-preconditioned by one V-cycle; CGS2 orthogonalisation with
// batched device dot products; Givens rotations on the host (one small D2H copy + sync per iteration).
// bnorm_known >= 0: the 2-norm of b, already on the host (the Newton driver's |F|: b = -F) - saves a reduction and its read-back
static int fgmres(pgx_handle* h, const double* b, double* x, const pgx_snes_opts* o, int* its_out, double* relres,
                  double bnorm_known = -1.0) {
  const size_t n2 = 2 * (size_t)h->nd;
  // Sharded: the Krylov space lives on OWNED dofs (basis vectors V_j, b, residuals are owned-compact, length nk, so the
  // tuned vector kernels run unchanged and every dot product is "local partial + one all-reduce"); the operators work
  // on local vectors with ghost rows (Z_j, x), with a gather after every SpMV.
  const bool dist = h->dist.on;
  const size_t nk = dist ? 2 * h->dist.nk_field() : n2;
  double* const wk = dist ? h->dist.wc : h->w;
  // P2: the two-level preconditioner is weaker on the late large-alpha systems (30-60 its): use the full basis
  const int m = (h->degree == 2) ? h->restart : std::min(std::max(o->ksp_restart, 1), h->restart);
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m);
  double bnorm = bnorm_known;
  int rc = PGX_OK;
  if (!(bnorm_known >= 0.0) && (rc = dev_norm(h, b, &bnorm, nk))) return rc;
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
  // Smoother damping.  The default 0.8 is the fast choice while psi is smooth on the mesh scale (lambda_max of the
  // Jacobi-scaled element matrices is 2); an overshot Newton iterate makes exp(psi) jump by orders of magnitude
  // inside single elements, lambda_max approaches its bound 3 (dofs per triangle) and only omega < 2/3 is a
  // convergent smoother.  FGMRES tolerates a changing preconditioner, so on stagnation (30 iterations that gain
  // < 10x) the rest of this Newton solve runs at the unconditionally stable value.
  const double omega_safe = 0.6;
  double omega = (h->omega_now > 0.0) ? std::min(h->omega_now, o->mg_omega) : o->mg_omega;
  while (true) {
    double beta;
    if (first_cycle && its >= o->ksp_max_it) break;
    if (first_cycle) {
      beta = bnorm;
      pgxk_scale_copy(h->st, nk, 1.0 / beta, b, h->V);
    } else {
      // r = b - J x
      PhaseTimer t(h, 3);
      spmv_dev(h, x, h->w);
      if (dist) gather_owned(h, h->w, wk);
      pgxk_scale_copy(h->st, nk, -1.0, wk, wk);
      pgxk_axpy(h->st, nk, 1.0, b, wk);
      rc = dev_norm(h, wk, &beta, nk);
      if (rc) return rc;
      res = beta;
      if (o->monitor > 1) printf("      ksp true residual after cycle: %.6e (rel %.3e)\n", beta, beta / bnorm);
      if (beta <= target) break;
      // attainable-accuracy exit: a full restart cycle that gains < 10x once we are at LU-level residuals
      if (beta > 0.1 * prev_cycle_res && beta <= 1e-7 * bnorm) break;
      if (its >= o->ksp_max_it) break;
      prev_cycle_res = beta;
      pgxk_scale_copy(h->st, nk, 1.0 / beta, wk, h->V);
    }
    first_cycle = false;
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    // Lazy normalisation (round 4): a new basis vector stays as the Gram-Schmidt pass left it, W_{j+1} = w', and only its scale
    // s_{j+1} = 1 / |w'| is kept (v_i = s_i W_i): the V-cycle multiplies its right-hand side by s_j as its first launch reads it, the
    // batched dot products come back as s_i^2 (W_i . w) - the coefficients the projection w' = w - sum_i (v_i . w) v_i applies to the
    // stored W_i - and the pass over w that only divided it by its norm (67 MB read + written per iteration) is gone.  Only where
    // the preconditioner is the single-precision cycle entered on level 0 (it is the one that takes the factor).
    const bool lazy = h->lazy_norm && h->cgs_selective && !h->lu_active && h->degree == 1 && m + 2 <= PGX_DOT_SCALE_MAX && !h->lev.empty() &&
                      f32_cycle_ok(h, 0, o->mg_nu);  // (the non-selective CGS2 kernels carry no scale factors: that mode runs un-lazy)
    // Z_j in single precision (round 5): on this path z_j leaves a float cycle, so storing it as one float2 field loses nothing - the
    // cycle's last launch writes it in place, the operator apply reads it (k_st_spmv_r<true>), the solution update sums it
    // (k_lincomb_f2): 100 MB less per Krylov iteration at 2048^2 than the fp64 pair.  w = J z_j, the basis V and H stay fp64.
    const bool zf32 = z_f32_active(h, o->mg_nu);
    float2* const Zf = reinterpret_cast<float2*>(h->Z);
    PgxDotScale sc2;  // s_i^2
    std::vector<double> sv((size_t)m + 2, 1.0);  // s_i
    for (int i = 0; i < PGX_DOT_SCALE_MAX; ++i) sc2.s[i] = 1.0;
    int j = 0;
    for (; j < m && its < o->ksp_max_it; ++j) {
      double* vj = h->V + (size_t)j * nk;
      double* zj = h->Z + (size_t)j * n2;
      {
        PhaseTimer t(h, 4);
        h->rhs_scale = lazy ? sv[j] : 1.0;
        h->zf_out = zf32 ? Zf + (size_t)j * h->nd : nullptr;
        rc = precond(h, vj, zj, o->mg_nu, omega);
        h->zf_out = nullptr;
        h->rhs_scale = 1.0;
        if (rc) return rc;
      }
      {
        PhaseTimer t(h, 3);
        if (dist) {
          spmv_dev(h, zj, h->w);
          gather_owned(h, h->w, h->V + (size_t)(j + 1) * nk);
        } else if (zf32) {
          double* const wn = h->V + (size_t)(j + 1) * n2;
          pgxk_st_spmv(h->st, h->lev[0], h->alpha, nullptr, nullptr, h->xcd_remap ? 1 : 0, wn, wn + h->nd, Zf + (size_t)j * h->nd);
        } else {
          spmv_dev(h, zj, h->V + (size_t)(j + 1) * n2);
        }
      }
      // w = J z_j was written straight into the V_{j+1} slot.  CGS2 (classical Gram-Schmidt, always two
      // passes: one pass loses orthogonality on these ill-conditioned systems and the true residual stalls).
      // Each pass is ONE batched dot kernel [h; ww] = [V_0..V_j, w]^T w plus one batched axpy; the norm of the
      // result comes from Pythagoras on the second pass (|w''|^2 = ww' - |h2|^2, cancellation-free because
      // the second pass removes almost nothing), so no separate norm kernel.
      double* wj = h->V + (size_t)(j + 1) * nk;
      double hn = 0.0;
      {
        PhaseTimer t(h, 5);
        double* d_h1 = h->d_small;
        double* d_h2 = h->d_small + (m + 2);
        const PgxDotScale* const dsc = lazy ? &sc2 : nullptr;
        // pass 1: h1 = V^T w.  passes 2+3 fused: w' = w - V h1 and [h2; |w'|^2] in one sweep over the basis.
        // Sharded: each batch of partial dot products is completed by ONE packed all-reduce, enqueued on the stream.
        pgxk_multidot(h->st, nk, j + 1, h->V, nk, wj, h->partials, d_h1, dsc);
        if (dist && (rc = allreduce_dev(h, d_h1, j + 1))) return rc;
        double wp2, h1h1 = 0.0, hh = 0.0;
        bool second;
        if (h->cgs_selective) {
          // lean second pass: w' = w - V h1 and |w'|^2; V^T w' (for the second projection) only if the test below asks for it
          pgxk_multiaxpy_norm(h->st, nk, j + 1, h->V, nk, d_h1, wj, h->partials, d_h2 + j + 1);
          if (dist && (rc = allreduce_dev(h, d_h2 + j + 1, 1))) return rc;
          if ((rc = replica_agree(h, h->d_small, 2 * (size_t)(m + 2)))) return rc;
          if ((rc = fetch_small(h, d_h1, (size_t)(j + 1), 0, d_h2 + j + 1, 1, (size_t)(m + 2) + j + 1))) return rc;
          for (int i = 0; i <= j; ++i) {
            h->h_small[i] /= sv[i];  // lazy: the device holds s_i^2 (W_i . w); the Hessenberg entry is s_i (W_i . w)
            h1h1 += h->h_small[i] * h->h_small[i];
          }
          wp2 = h->h_small[(m + 2) + j + 1];
          // "twice is enough" (Kahan / Parlett; Daniel-Gragg-Kaufman-Stewart): the second projection is only needed when the
          // first one cancelled most of w, |w'| < eta |w| with |w|^2 = |w'|^2 + |h1|^2
          second = wp2 < h->cgs_eta2 * (wp2 + h1h1);
          if (second) {
            pgxk_multidot(h->st, nk, j + 1, h->V, nk, wj, h->partials, d_h2, dsc);
            if (dist && (rc = allreduce_dev(h, d_h2, j + 1))) return rc;
            if ((rc = replica_agree(h, d_h2, (size_t)(j + 1)))) return rc;
            if ((rc = fetch_small(h, d_h2, (size_t)(j + 1), (size_t)(m + 2)))) return rc;
            for (int i = 0; i <= j; ++i) h->h_small[(m + 2) + i] /= sv[i];
          }
        } else {
          if (j + 1 <= 60) {
            pgxk_axpy_dot(h->st, nk, j + 1, h->V, nk, d_h1, wj, h->partials2, d_h2);
          } else {  // beyond the fused kernel's LDS capacity (61 slices of 2 KB): two separate passes
            pgxk_multiaxpy(h->st, nk, j + 1, h->V, nk, d_h1, wj);
            pgxk_multidot(h->st, nk, j + 2, h->V, nk, wj, h->partials, d_h2);
          }
          if (dist && (rc = allreduce_dev(h, d_h2, j + 2))) return rc;
          if ((rc = replica_agree(h, h->d_small, 2 * (size_t)(m + 2)))) return rc;
          if ((rc = fetch_small(h, h->d_small, 2 * (size_t)(m + 2)))) return rc;
          wp2 = h->h_small[(m + 2) + j + 1];
          second = true;
        }
        for (int i = 0; i <= j; ++i) {
          const double h2 = second ? h->h_small[(m + 2) + i] : 0.0;
          H[(size_t)i * m + j] = h->h_small[i] + h2;
          hh += h2 * h2;
        }
        if (second) {
          hn = std::sqrt(std::max(wp2 - hh, 0.0));
          // last pass fused with the normalisation: v_{j+1} = (w' - V h2) / hn   (|w''|^2 = |w'|^2 - |h2|^2, Pythagoras)
          if (hn > 0.0) {
            if (lazy)
              pgxk_multiaxpy(h->st, nk, j + 1, h->V, nk, d_h2, wj);
            else
              pgxk_multiaxpy_scale(h->st, nk, j + 1, h->V, nk, d_h2, 1.0 / hn, wj);
          }
        } else {
          hn = std::sqrt(std::max(wp2, 0.0));
          if (hn > 0.0 && !lazy) pgxk_scale_copy(h->st, nk, 1.0 / hn, wj, wj);
          ++h->cgs_skipped;
        }
        if (lazy && hn > 0.0) {
          sv[j + 1] = 1.0 / hn;
          sc2.s[j + 1] = sv[j + 1] * sv[j + 1];
        }
        ++h->cgs_total;
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
      if (dist) ++h->dist.n_krylov;
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
      // Stagnation test.  A healthy solve gains a factor 5-8 per iteration; a smoother that diverges on the rough late iterates
      // shows within a dozen iterations.  FGMRES tolerates a changing preconditioner, so the damping is switched IN PLACE (no
      // restart: the basis built so far stays useful) as soon as `stag_its` iterations have gained less than 1e-4 (round 1
      // waited for a full cycle of 30 that gained < 10x and restarted).
      if (omega > omega_safe && j + 1 == std::min(m, h->stag_its) && res > h->stag_gain * beta) {
        h->omega_now = omega = omega_safe;  // sticky until pgx_newton_solve returns: later iterates are rough too
        if (o->monitor > 1) printf("      ksp stagnates: smoother damping -> %.2f for the rest of this Newton solve\n", omega);
      }
    }
    // y = H^-1 g (upper triangular), x += Z y
    for (int i = j - 1; i >= 0; --i) {
      double s = g[i];
      for (int k = i + 1; k < j; ++k) s -= H[(size_t)i * m + k] * y[k];
      y[i] = s / H[(size_t)i * m + i];
    }
    for (int i = 0; i < j; ++i) h->h_small[i] = y[i];
    HIPCHK(hipMemcpyAsync(h->d_small, h->h_small, sizeof(double) * j, hipMemcpyHostToDevice, h->st));
    if (zf32)
      pgxk_lincomb_f2(h->st, (size_t)h->nd, j, Zf, (size_t)h->nd, h->d_small, x, x + h->nd);
    else
      pgxk_lincomb(h->st, n2, j, h->Z, n2, h->d_small, x, 1);
    // (no synchronisation: the loop head recomputes the TRUE residual b - Jx - it decides convergence, not the Arnoldi estimate - and
    // its read-back is ordered behind the upload of y on the stream, so h_small is not touched again before that copy is done)
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
    int rc;
    if (h->dist.on) {  // collective: 2-norm over the owned entries of all ranks
      gather_owned(h, h->F, h->dist.wc);
      rc = dev_norm(h, h->dist.wc, fnorm, 2 * h->dist.nk_field());
    } else {
      rc = dev_norm(h, h->F, fnorm);
    }
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
  {
    const int rc = jacobian_dev(h, xd);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(h->st));
  HIPCHK(hipGetLastError());
  return PGX_OK;
}

extern "C" int pgx_csr_export(pgx_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* K,
                              double* M, double* D) {
  NEED(h);
  const std::vector<int32_t>& hr = (h->degree == 2) ? h->s_h_rowptr : h->h_rowptr;
  const std::vector<int32_t>& hc = (h->degree == 2) ? h->s_h_col : h->h_col;
  if (nrows) *nrows = h->nd;
  if (nnz) *nnz = h->s_nnz;
  if (rowptr) memcpy(rowptr, hr.data(), sizeof(int32_t) * (h->nd + 1));
  if (col) memcpy(col, hc.data(), sizeof(int32_t) * h->s_nnz);
  if (K) HIPCHK(hipMemcpy(K, h->s_K, sizeof(double) * h->s_nnz, hipMemcpyDeviceToHost));
  if (M) HIPCHK(hipMemcpy(M, h->s_M, sizeof(double) * h->s_nnz, hipMemcpyDeviceToHost));
  if (D) {
    if (!h->jac_valid) {
      h->err = "pgx_csr_export(D) before pgx_jacobian_fill";
      return PGX_ESTATE;
    }
    HIPCHK(hipMemcpy(D, h->s_D, sizeof(double) * h->s_nnz, hipMemcpyDeviceToHost));
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
  if (h->dist.on && (rc = halo_solution(h, h->V, h->V + h->nd))) return rc;  // ghost entries from their owners
  spmv_dev(h, h->V, h->w);
  return copy_out(h, y, h->w);  // owned rows are J x of the GLOBAL operator; ghost rows are not meaningful
}

extern "C" int pgx_smoother_bench(pgx_handle* h, int reps, double* avg_ms, double* bytes) {
  NEED(h);
  if (reps < 1 || !avg_ms) return PGX_EINVAL;
  if (!h->jac_valid || !h->structured || h->lev.size() < 2 || h->dist.on) {
    h->err = "pgx_smoother_bench needs a structured single-GPU handle with a grid hierarchy and a filled Jacobian";
    return PGX_ESTATE;
  }
  if (h->degree == 2) {
    // P2: the time-dominant kernel is the patch sweep (k_patch_apply + k_patch_edges) on the current Jacobian's inverses
    if (!h->patch_nn || !h->p2_patch) {
      h->err = "pgx_smoother_bench (P2): no patch smoother on this handle";
      return PGX_ESTATE;
    }
    const int rcp = ensure_patches(h);
    if (rcp) return rcp;
    const int nd = h->nd, nv = h->n, NN = h->patch_nn, P = 2 * NN;
    pgxk_set(h->st, 2 * (size_t)nd, 1.0, h->rhs);
    HIPCHK(hipMemsetAsync(h->w, 0, sizeof(double) * 2 * nd, h->st));
    auto sweep = [&]() {
      pgxk_patch_sweep(h->st, nv, NN, nv, nd, h->pdof, h->edge_ends, h->pinv, h->patch_f32, h->patch_sym, h->rhs, h->rhs + nd, h->patch_omega,
                       h->w, h->w + nd, h->p2_su, h->p2_sp);
    };
    for (int k = 0; k < 3; ++k) sweep();
    HIPCHK(hipEventRecord(h->e0, h->st));
    for (int k = 0; k < reps; ++k) sweep();
    HIPCHK(hipEventRecord(h->e1, h->st));
    HIPCHK(hipEventSynchronize(h->e1));
    float msp = 0;
    HIPCHK(hipEventElapsedTime(&msp, h->e0, h->e1));
    *avg_ms = (double)msp / reps;
    if (bytes) {
      // per patch: the inverse, the dof table, the residual at its 2 NN dofs (read once: every dof is gathered by its patches out
      // of L2, HBM sees each value once -> 16 B per dof), the vertex iterate read + written, the parked edge contributions; per edge
      // (k_patch_edges): the two parked pairs read, the edge iterate read + written
      const double inv = (double)pgxk_patch_inverse_bytes(nv, NN, h->patch_f32, h->patch_sym);
      const double ne = (double)(nd - nv);
      *bytes = inv + 4.0 * NN * nv + 16.0 * nd + 32.0 * nv + 2.0 * 2.0 * 4.0 * ne + (16.0 + 32.0) * ne;
    }
    (void)P;
    return PGX_OK;
  }
  GridLevel& L = h->lev[0];
  GridLevel& C = h->lev[1];
  const size_t n = (size_t)L.n;
  // right-hand side and iterate: any finite data (the kernel's cost does not depend on the values); coarse correction zero
  pgxk_set(h->st, 2 * n, 1.0, h->rhs);
  pgxk_set(h->st, 2 * n, 0.5, h->w);
  HIPCHK(hipMemsetAsync(C.xu, 0, sizeof(double) * C.n, h->st));
  HIPCHK(hipMemsetAsync(C.xp, 0, sizeof(double) * C.n, h->st));
  const int remap = h->xcd_remap ? 1 : 0;
  const bool f32 = f32_cycle_ok(h, 0, 6);
  if (f32) {  // the single-precision launch of the same role: 3 sweeps on x + P x_c, float2 in and out
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)L.bf, 0x3f800000, 2 * n, h->st));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)L.xf, 0x3f000000, 2 * n, h->st));
    if (C.f32) HIPCHK(hipMemsetAsync(C.xf, 0, sizeof(float2) * C.n, h->st));
  }
  auto run = [&]() {
    if (f32)
      pgxk_f_smooth(h->st, 3, 0, L, h->alpha, L.xf, nullptr, nullptr, &C, C.f32 ? C.xf : nullptr, C.f32 ? nullptr : C.xu,
                    C.f32 ? nullptr : C.xp, 0.8, remap, L.xf2, nullptr, nullptr);
    else
      pgxk_st_smoothK(h->st, 3, 1, L, h->alpha, h->w, h->w + n, &C, C.xu, C.xp, h->rhs, h->rhs + n, 0.8, remap, h->tmp_u, h->tmp_p);
  };
  for (int k = 0; k < 3; ++k) run();
  HIPCHK(hipEventRecord(h->e0, h->st));
  for (int k = 0; k < reps; ++k) run();
  HIPCHK(hipEventRecord(h->e1, h->st));
  HIPCHK(hipEventSynchronize(h->e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->e0, h->e1));
  *avg_ms = (double)ms / reps;
  if (bytes)  // D stencil + right-hand side + iterate + coarse correction + result
    *bytes = f32 ? (16.0 * n + 8.0 * n + 8.0 * n + (C.f32 ? 8.0 : 16.0) * C.n + 8.0 * n) : 8.0 * (4.0 * n + 2.0 * n + 2.0 * n + 2.0 * C.n + 2.0 * n);
  return PGX_OK;
}

extern "C" int pgx_vcycle_bench(pgx_handle* h, int level, int reps, double* avg_ms, int* n_level) {
  NEED(h);
  if (reps < 1 || !avg_ms) return PGX_EINVAL;
  if (!h->jac_valid || !h->structured || h->lev.size() < 2 || h->dist.on || h->degree != 1) {
    h->err = "pgx_vcycle_bench needs a structured single-GPU P1 handle with a grid hierarchy and a filled Jacobian";
    return PGX_ESTATE;
  }
  if (level < 0) level = h->tail_start > 0 ? h->tail_start : (int)h->lev.size() - 1;
  if (level >= (int)h->lev.size()) return PGX_EINVAL;
  pgx_snes_opts od;
  pgx_default_opts(&od);
  GridLevel& L = h->lev[level];
  double *bu = level == 0 ? h->rhs : L.bu, *bp = level == 0 ? h->rhs + L.n : L.bp;
  double *xu = level == 0 ? h->w : L.xu, *xp = level == 0 ? h->w + L.n : L.xp;
  pgxk_set(h->st, (size_t)L.n, 1.0, bu);
  pgxk_set(h->st, (size_t)L.n, 1.0, bp);
  const bool f32 = level > 0 && f32_cycle_ok(h, level, od.mg_nu);  // a single-precision level below the finest: float2 right-hand side
  if (f32) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)L.bf, 0x3f800000, 2 * (size_t)L.n, h->st));
  auto run = [&]() {
    if (f32)
      vcycle_f(h, level, nullptr, nullptr, nullptr, nullptr, od.mg_nu, od.mg_omega);
    else
      vcycle(h, level, bu, bp, xu, xp, od.mg_nu, od.mg_omega);
  };
  for (int k = 0; k < 3; ++k) run();
  HIPCHK(hipEventRecord(h->e0, h->st));
  for (int k = 0; k < reps; ++k) run();
  HIPCHK(hipEventRecord(h->e1, h->st));
  HIPCHK(hipEventSynchronize(h->e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->e0, h->e1));
  *avg_ms = (double)ms / reps;
  if (n_level) *n_level = L.n;
  return PGX_OK;
}

static int spmv_bench_impl(pgx_handle* h, int reps, double* avg_ms, double* bytes, bool cold) {
  NEED(h);
  if (reps < 1 || !avg_ms) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_spmv_bench before pgx_jacobian_fill";
    return PGX_ESTATE;
  }
  const size_t n2 = 2 * (size_t)h->nd;
  pgxk_set(h->st, n2, 1.0, h->V);
  for (int k = 0; k < 3; ++k) spmv_dev(h, h->V, h->w);
  // Launches in the cache state of a solve, not back to back: the matrix-free kernel's whole footprint (273 MB at 2048^2) would
  // otherwise sit in the 256 MB Infinity Cache (42 us per launch in a tight loop, 49-50 us inside the solves per rocprofv3).
  // Multigrid handles replay the solver's own sequence - one V-cycle producing z, then J z, exactly as in an FGMRES iteration -
  // and subtract the time of the V-cycles alone; other handles sweep 512 MB of idle storage (a dot product) between two
  // applies.  Each batch sits between ONE pair of events: a pair around a single 50 us kernel would add the ~50 us of its two
  // barrier packets.
  // cold: every apply follows a 512 MB sweep of unrelated storage, so neither x nor the stencils sit in the 256 MB Infinity Cache
  const bool mg = !cold && h->structured && h->degree == 1 && !h->dist.on && !h->lu_active && h->lev.size() > 1;
  pgx_snes_opts od;
  pgx_default_opts(&od);
  const bool zf = mg && z_f32_active(h, od.mg_nu);
  const size_t flush = std::min<size_t>((size_t)h->restart * n2, ((size_t)512 << 20) / sizeof(double));
  // nine rounds of (batch with applies, batch without), median of the differences: clock and power drift between two ~50 ms
  // batches is of the order of the quantity measured
  std::vector<double> diff;
  for (int round = 0; round < 9; ++round) {
    double t[2] = {0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) {
      HIPCHK(hipEventRecord(h->e0, h->st));
      for (int k = 0; k < reps; ++k) {
        if (mg) {
          h->zf_out = zf ? reinterpret_cast<float2*>(h->Z) : nullptr;  // as in fgmres: the cycle leaves z as ONE float2 field ...
          const int rc = precond(h, h->V, h->Z, od.mg_nu, od.mg_omega);
          h->zf_out = nullptr;
          if (rc) return rc;
        } else {
          pgxk_multidot(h->st, flush, 1, h->Z, 0, h->Z, h->partials, h->d_small);
        }
        if (pass == 0) {
          if (zf)  // ... and the apply reads it (k_st_spmv_r<true>)
            pgxk_st_spmv(h->st, h->lev[0], h->alpha, nullptr, nullptr, h->xcd_remap ? 1 : 0, h->w, h->w + h->nd, reinterpret_cast<const float2*>(h->Z));
          else
            spmv_dev(h, mg ? h->Z : h->V, h->w);
        }
      }
      HIPCHK(hipEventRecord(h->e1, h->st));
      HIPCHK(hipEventSynchronize(h->e1));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, h->e0, h->e1));
      t[pass] = ms;
    }
    diff.push_back((t[0] - t[1]) / reps);
  }
  std::sort(diff.begin(), diff.end());
  *avg_ms = diff[diff.size() / 2];
  if (bytes) {
    if (h->spmv_stencil && h->structured && h->degree == 1)
      // matrix-free: x read once (16 B per vertex; 8 B as the float2 field of an FGMRES z_j), y written (16 B), half-stored D stencil
      // (4 x 8 B), Dirichlet mask (1 B); K and M are seven constants each
      *bytes = ((zf ? 8.0 : 16.0) + 16.0 + 4.0 * sizeof(dsten_t) + 1.0) * h->nd;
    else if (h->degree == 2 && h->p2st.state == 2 && h->p2st.select && (!h->dist.on || h->p2st.dist_enable) && h->spmv_stream && h->spmv_bal) {
      // structured P2 apply: 46 D values per interior group (SoA copy), nothing else of the matrix; the frame rows in CSR form
      // (column + K + M + D = 28 B per entry, the row list); x read once; y written
      const double nfast = (double)h->p2st.ni * h->p2st.nj;
      *bytes = 368.0 * nfast + 28.0 * ((double)h->s_nnz - 46.0 * nfast) + 4.0 * h->p2st.nframe + 8.0 * n2 + 8.0 * n2;
    } else if (h->s_code && h->spmv_bal && h->s_blk && h->spmv_stream)
      // (K, M) dictionary: column (4 B) + code (1 B) + D (8 B) per scalar nnz; rowptr; x read once; y written
      *bytes = 13.0 * h->s_nnz + 4.0 * (h->nd + 1) + 8.0 * n2 + 8.0 * n2;
    else  // one pattern (4 B) + three value streams (24 B) per scalar nnz; rowptr; x read once; y written
      *bytes = 28.0 * h->s_nnz + 4.0 * (h->nd + 1) + 8.0 * n2 + 8.0 * n2;
  }
  return PGX_OK;
}
extern "C" int pgx_spmv_bench(pgx_handle* h, int reps, double* avg_ms, double* bytes) {
  return spmv_bench_impl(h, reps, avg_ms, bytes, false);
}
extern "C" int pgx_spmv_bench_cold(pgx_handle* h, int reps, double* avg_ms, double* bytes) {
  return spmv_bench_impl(h, reps, avg_ms, bytes, true);
}

extern "C" int pgx_comm_counts(pgx_handle* h, int64_t out[4], int reset) {
  NEED(h);
  if (!out) return PGX_EINVAL;
  out[0] = h->dist.n_halo;
  out[1] = h->dist.n_allreduce;
  out[2] = h->dist.n_vcycle;
  out[3] = h->dist.n_krylov;
  if (reset) h->dist.n_halo = h->dist.n_allreduce = h->dist.n_vcycle = h->dist.n_krylov = 0;
  return PGX_OK;
}

extern "C" int pgx_p2_stencil_info(pgx_handle* h, int32_t out[5]) {
  NEED(h);
  if (!out) return PGX_EINVAL;
  const pgx_handle::P2St& S = h->p2st;
  out[0] = (h->degree == 2 && (!h->dist.on || S.dist_enable)) ? S.state : 0;
  out[1] = S.i0, out[2] = S.ni, out[3] = S.j0, out[4] = S.nj;
  return PGX_OK;
}
extern "C" int pgx_spmv_select(pgx_handle* h, int kind, int* active) {
  NEED(h);
  if (h->degree == 2) {  // P2: 3 (or 1) = the structured apply of pgx_p2st.hip where the mesh allows it, 0 = the block-CSR kernel
    if (kind == 0) h->p2st.select = 0;
    else if (kind == 1 || kind == 3) h->p2st.select = 1;
    else if (kind != -1) return PGX_EINVAL;
    if (active) *active = (h->p2st.state && h->p2st.select && (!h->dist.on || h->p2st.dist_enable) && h->spmv_stream && h->spmv_bal) ? 3 : 0;
    return PGX_OK;
  }
  if (kind == 0 || kind == 1 || kind == 2) h->spmv_stencil = kind;
  else if (kind != -1) return PGX_EINVAL;
  if (active) *active = (h->spmv_stencil && h->structured && h->degree == 1) ? h->spmv_stencil : 0;
  return PGX_OK;
}

extern "C" int pgx_observables(pgx_handle* h, double out[6]) {
  NEED(h);
  if (!out) return PGX_EINVAL;
  {
    PhaseTimer t(h, 6);
    if (h->degree == 2 && !h->dist.on) {
      pgxk_observables_p2_cells(h->st, h->nc, h->nd, h->cdofs, h->coords, h->x, h->xk, h->alpha, h->f, h->q2,
                                h->obs_partials, h->obs_blocks, h->geoq);
      pgxk_observables_final(h->st, h->obs_blocks, h->obs_partials, h->d_out6);
    } else if (h->dist.on && h->degree == 2) {
      const Dist& D = h->dist;  // P2 on a strip: the owned cells' partial sums, one packed all-reduce (raw sums, as below)
      pgxk_observables_p2_cells(h->st, D.ncell_own, h->nd, h->cdofs + 6 * (size_t)D.cell0, h->coords, h->x, h->xk, h->alpha, h->f,
                                h->q2, h->obs_partials, pgxk_observables_blocks(D.ncell_own));
      pgxk_observables_final_raw(h->st, pgxk_observables_blocks(D.ncell_own), h->obs_partials, h->d_out6);
      const int rc = allreduce_dev(h, h->d_out6, 6);
      if (rc) return rc;
    } else if (h->dist.on) {
      // owned cells only (a contiguous range in row-major cell order), raw sums, ONE packed all-reduce for the six
      // scalars (the reference all-reduces each one separately, obstacle_pg.py:196-201), then abs / sqrt
      const Dist& D = h->dist;
      pgxk_observables(h->st, D.ncell_own, h->n, h->cells + 3 * (size_t)D.cell0, h->coords, h->x, h->xk, h->alpha, h->f,
                       h->q, h->obs_partials, pgxk_observables_blocks(D.ncell_own), h->d_out6, 1);
      const int rc = allreduce_dev(h, h->d_out6, 6);
      if (rc) return rc;
    } else {
      pgxk_observables(h->st, h->nc, h->n, h->cells, h->coords, h->x, h->xk, h->alpha, h->f, h->q, h->obs_partials,
                       h->obs_blocks, h->d_out6, 0, h->geoq);
    }
  }
  {
    int rcb = replica_check(h, h->d_out6, 6, "the observables");  // fixed-shape reductions: identical on every replica
    if (!rcb) rcb = replica_agree(h, h->d_out6, 6);                // the stopping test of the proximal loop reads these
    if (rcb) return rcb;
  }
  {
    const int rcf = fetch_small(h, h->d_out6, 6);
    if (rcf) return rcf;
  }
  for (int k = 0; k < 6; ++k) out[k] = h->h_small[k];
  if (h->dist.on) {
    out[1] = std::fabs(out[1]);
    out[4] = std::sqrt(out[4]);
    out[5] = std::sqrt(out[5]);
  }
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
  pgx_snes_opts optv = *opts;
  // auto: chosen so that the final primal field stays within 1e-10 of the LU oracle on EVERY case measured
  // (tools/lu_accuracy_check.py, tools/tolerance_sweep.py): small / unstructured meshes are the sensitive ones
  if (!(optv.ksp_rtol > 0.0)) optv.ksp_rtol = (h->degree == 2) ? 1e-11 : 1e-10;
  opts = &optv;
  // pc_type: 0 auto (geometric multigrid for P1 on a structured mesh; sparse LU for P2, whose two-level cycle is not robust
  // on the late large-alpha systems, and for general meshes, which have no grid hierarchy - e.g. the reference's own gmsh
  // disk, obstacle_pg.py:64-65), 1 multigrid V-cycle, 2 sparse LU (what the reference asks PETSc/MUMPS for)
  // P2 on a structured mesh (round 3): the two-level cycle with the vertex-star patch smoother first (pcycle_p2_patch: 8-20
  // iterations per Newton step at every size, a fifth of a factorisation's time) and the sparse LU only for a Newton solve in
  // which that cycle stagnates - the overshot iterates of settings B's large alpha jumps, where exp(psi) is rough on the mesh
  // scale; the LU then stays in place for the rest of that pgx_newton_solve call.
  const bool mg_first = optv.pc_type == 0 && h->degree == 2 && h->structured && h->lev.size() > 1 && h->patch_nn && h->p2_patch &&
                        !h->dist.on && !h->lu_comm;
  bool use_lu = h->lu_comm || optv.pc_type == 2 ||
                (optv.pc_type == 0 && (h->degree == 2 || !h->structured) && !h->dist.on && !mg_first);
  h->lu_active = false;
  if (use_lu) {
    int rcl = ensure_lu(h);
    if (rcl) return rcl;
  }
  // Sharded P2 has no sparse-LU rescue (the factorisation is not sharded): where the single handle would switch to it - the
  // overshot iterates of settings B, on which the patch cycle needs a few hundred iterations instead of ten - the sharded solve
  // keeps iterating on its un-restarted basis (up to 300 vectors).  512^2 settings B on 4 strips: the golden's Newton counts
  // 5,4,3,2,1,1,4,1 (tests/test_gpu_sharded.py).
  // sharded P2 has no sparse-LU fallback for a stagnating two-level cycle: it runs the Krylov method longer instead - but only where
  // the caller left the default (an explicit ksp_max_it is the caller's decision; include/pgx.h)
  if (h->dist.on && h->degree == 2 && optv.ksp_max_it == 200) optv.ksp_max_it = 400;
  const size_t n2 = 2 * (size_t)h->nd;
  PgxSolveScope scope(h->st, h->prof, &h->lu_active);
  // EVERY way out of the Newton loop - also the early `return rc` of a failed Krylov solve, halo exchange or replica check - leaves
  // a hierarchy whose finest D(psi) was refreshed in passing (and, lean, CSR rows that were not): no export / CSR product of a
  // matrix that mixes iterates afterwards
  struct JacStale {
    pgx_handle* h;
    ~JacStale() {
      if (h->dh_interior || h->dv_lean) h->jac_valid = false;
    }
  } jac_stale{h};
  PgxRange range("pgx:newton_solve");
  int its = 0, lin = 0, rsn = 0;
  double fnorm = 0, fnorm0 = 0, ttol = 0;
  h->omega_now = 0.0;
  int rc = PGX_OK;
  // Sharded: x, xw, dx, F are local vectors (owned + ghost rows); norms run on owned-compact copies (rhs doubles as the
  // compact copy of F), and the ghost rows of the iterate are refreshed after every update, before the next assembly.
  const bool dist = h->dist.on;
  const size_t nk = dist ? 2 * h->dist.nk_field() : n2;
  auto owned_norm = [&](const double* v, double* out) -> int {
    if (!dist) return dev_norm(h, v, out);
    gather_owned(h, v, h->dist.wc);
    return dev_norm(h, h->dist.wc, out, nk);
  };
  // matrix-free operator + multigrid preconditioner: nobody reads the CSR form of D inside this solve (see residual_dev)
  const int with_d = (!use_lu && !mg_first && h->degree == 1 && h->structured && h->spmv_stencil == 1 && h->lean_d) ? 2 : 1;
  pgxk_scale_copy(h->st, n2, 1.0, h->x, h->xw);
  residual_dev(h, h->xw, h->F, with_d);
  if ((rc = replica_check(h, h->F, n2, "the residual"))) return rc;
  if (dist) {
    gather_owned(h, h->F, h->rhs);
    rc = dev_norm(h, h->rhs, &fnorm, nk);
  } else {
    rc = dev_norm(h, h->F, &fnorm);
  }
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
    rc = jacobian_dev(h, h->xw, true);
    if (rc) return rc;
    if (use_lu) {
      if ((rc = lu_factor(h))) return rc;
      h->lu_active = true;
    }
    pgxk_scale_copy(h->st, nk, -1.0, dist ? h->rhs : h->F, h->rhs);
    int kits = 0;
    double relres = 0;
    if (mg_first && !use_lu) {
      pgx_snes_opts ol = optv;
      ol.ksp_max_it = std::min(optv.ksp_max_it, h->p2_fallback_its);
      rc = fgmres(h, h->rhs, h->dx, &ol, &kits, &relres, fnorm);
      if (rc) return rc;
      // stagnation = the iteration cap was reached without convergence (an early exit at the attainable accuracy, a few 1e-10
      // after 10-15 iterations on the late systems, is not): factorise, and keep the factorisation for this solve
      if (kits >= ol.ksp_max_it && !(relres <= 10.0 * optv.ksp_rtol)) {
        if (opts->monitor) printf("    KSP (patch multigrid) %d its, rel residual %.3e: sparse LU for the rest of this solve\n", kits, relres);
        lin += kits;
        if ((rc = ensure_lu(h))) return rc;
        if ((rc = lu_factor(h))) return rc;
        h->lu_active = true;
        use_lu = true;
        ++h->p2_fallbacks;
        rc = fgmres(h, h->rhs, h->dx, opts, &kits, &relres, fnorm);
        if (rc) return rc;
      }
    } else {
      rc = fgmres(h, h->rhs, h->dx, opts, &kits, &relres, fnorm);
      if (rc) return rc;
    }
    lin += kits;
    ++its;
    if (opts->monitor) printf("    KSP its %d  rel residual %.3e\n", kits, relres);
    if (!(relres <= std::max(opts->ksp_rtol, 1e-7)) || !std::isfinite(relres)) {
      rsn = PGX_SNES_DIVERGED_LINEAR_SOLVE;
      break;
    }
    pgxk_axpy(h->st, n2, 1.0, h->dx, h->xw);
    if (dist && (rc = halo_solution(h, h->xw, h->xw + h->nd))) return rc;
    residual_dev(h, h->xw, h->F, with_d);
    if ((rc = replica_check(h, h->F, n2, "the residual"))) return rc;
    if (dist) {
      gather_owned(h, h->F, h->rhs);
      rc = dev_norm(h, h->rhs, &fnorm, nk);
    } else {
      rc = dev_norm(h, h->F, &fnorm);
    }
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
      rc = owned_norm(h->dx, &snorm);
      if (rc) return rc;
      rc = owned_norm(h->xw, &xnorm);
      if (rc) return rc;
      if (snorm < opts->snes_stol * xnorm)
        rsn = PGX_SNES_CONVERGED_SNORM_RELATIVE;
      else if (fnorm > opts->snes_divtol * fnorm0)
        rsn = PGX_SNES_DIVERGED_DTOL;
    }
  }
  h->lu_active = false;
  if (rsn > 0) pgxk_scale_copy(h->st, n2, 1.0, h->xw, h->x);
  // the last residual evaluation refreshed D(psi) at the final iterate on the finest level only (k_resid_fill_grid writes
  // the interior rows of its stencil in passing): the hierarchy no longer describes ONE matrix until the next fill
  if (h->dh_interior || h->dv_lean) h->jac_valid = false;
  HIPCHK(hipStreamSynchronize(h->st));
  HIPCHK(hipGetLastError());  // a failed kernel launch anywhere in the solve must not pass silently
  if (h->prof) h->ms[7] += scope.stop();
  *reason = rsn;
  if (its_out) *its_out = its;
  if (lin_out) *lin_out = lin;
  return PGX_OK;
}
