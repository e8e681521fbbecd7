// pgx_mixed.h - what the example-06 and example-02 handles share: a mixed CSR Newton matrix on the device, state vectors,
// fixed-shape reductions, the SNES-mirroring Newton driver (newtonls, linesearch none) and its linear solve = sparse LU
// (pgx_nd) + iterative refinement on the exact operator.  Included by pgx_gc.hip and pgx_sg.hip (static: one copy per TU).
//
// Reference for the driver: PETSc SNES newtonls with `snes_linesearch_type none` as configured at
// examples/06_gradient_constraints/gradient_constraint_dolfinx.py:116-131 and examples/02_signorini/
// signorini_dolfinx.py:271-291,331-335; callback contract src/lvpp/problem.py:54-77,114-124.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pgx.h"
#include "../../include/pgx_nd.h"
#include "pgx_comm.h"
#include "pgx_scope.h"

struct MixedBase {
  int device = 0;
  hipStream_t st = nullptr;
  std::string err;
  int64_t ntot = 0, nnz = 0;
  int32_t *rowptr = nullptr, *col = nullptr;  // mixed CSR pattern (device)
  double* Jv = nullptr;                        // values of the current Jacobian
  double *x = nullptr, *xk = nullptr, *F = nullptr, *dx = nullptr, *xw = nullptr, *rhs = nullptr, *r = nullptr, *z = nullptr;
  double *partials = nullptr, *d_out = nullptr;
  double* h_out = nullptr;  // pinned
  std::vector<int32_t> h_rowptr, h_col;
  pgx_nd* lu = nullptr;
  double* gm_V = nullptr;  // Krylov basis of the GMRES safeguard (allocated on first use)
  int gm_m = 0;
  // Lazy refactorisation (EXPERIMENT, off: PGX_LAZY_LU=1 enables): from the second Newton step of a solve on, the factorisation
  // of the EARLIER iterate first serves as preconditioner of GMRES on the current exact Jacobian; only if that does not reach
  // the linear tolerance within `lazy_budget` iterations is the matrix factorised again.  Measured (tools/lazy_lu_ab.py): it
  // NEVER pays on these problems - 0 of 36 stale attempts converged within 10 iterations on example 06 at 1024^2 (13.8 -> 16.8 s),
  // 0 of 3 on example 02 at 70^3: between two Newton iterates the latent block N(psi) / D(psi) moves by orders of magnitude
  // where the constraint switches, and the 1e-12 true-residual bar leaves a stale factorisation no room.
  // Symmetrisation for the sparse LU (round 5): rows >= lu_flip_from of the Newton matrix are NEGATED on the way into pgx_nd_factor and
  // the same rows of every right-hand side on the way into pgx_nd_solve - x = J^-1 b = (S J)^-1 (S b), S = diag(I, -I).  Example 02's
  // [[alpha A, M_G^T], [-M_G, D]] becomes the symmetric [[alpha A, M_G^T], [M_G, -D]], which the LU factorises at half the flops
  // (pgx_nd_set_symmetric).  -1: off.  The rows are the last ones of the CSR, so their values are one contiguous range.
  double refine_eta = 1.0e-16;  // stop refining at this normwise backward error: the unit roundoff (PGX_MX_REFINE_ETA; 0 = never stop on it)
  int64_t lu_flip_from = -1;
  double* lu_flip_buf = nullptr;
  int lazy_lu = 0, lazy_budget = 10;
  bool lu_factored = false, stale_failed = false;
  long lazy_hits = 0, lazy_misses = 0, lazy_its = 0;
  // distributed handles (one per GPU, replicated iterate, distributed LU): every scalar that steers control flow - norms,
  // dot products of the line search - is taken from rank 0, so that all ranks make the same collective calls even though
  // their redundantly assembled residuals differ in the last bits (atomics)
  pgx_comm* comm = nullptr;
  bool jac_valid = false;
  bool prof = false;
  double ms[6] = {0, 0, 0, 0, 0, 0};  // [0] residual [1] jacobian [2] LU factor [3] LU solves [4] spmv [5] Newton total
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<void*> allocs;
  virtual void residual_dev(const double* xin, double* Fout) = 0;  // F(x) incl. the BC rows, asynchronous on st
  virtual void jacobian_dev(const double* xin) = 0;                // fills Jv at x
  virtual ~MixedBase() {}
};

#define MXHIP(call)                                               \
  do {                                                            \
    hipError_t e_ = (call);                                       \
    if (e_ != hipSuccess) {                                       \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return PGX_EHIP;                                            \
    }                                                             \
  } while (0)

template <typename T>
static int mx_alloc(MixedBase* h, T** p, size_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
    h->err = "hipMalloc of " + std::to_string(count * sizeof(T)) + " bytes failed";
    return PGX_ENOMEM;
  }
  h->allocs.push_back(q);
  *p = (T*)q;
  return PGX_OK;
}
#define MXALLOC(p, count)                        \
  do {                                           \
    int rc_ = mx_alloc(h, &(p), (size_t)(count)); \
    if (rc_) return rc_;                         \
  } while (0)

struct MxTimer {
  MixedBase* h;
  int slot;
  MxTimer(MixedBase* h_, int s) : h(h_), slot(s) {
    static const char* const names[6] = {"pgx:residual", "pgx:jacobian", "pgx:lu_factor", "pgx:lu_solve", "pgx:spmv", "pgx:newton"};
    pgx_roctx(names[s < 0 || s > 5 ? 5 : s]);
    if (h->prof) hipEventRecord(h->e0, h->st);
  }
  ~MxTimer() {
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

// y = A x (ABS: y = |A| |x|), 16 lanes per row
template <bool ABS>
static __global__ __launch_bounds__(256) void k_mx_spmv_t(int64_t nrows, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, const double* __restrict__ vals,
                                                          const double* __restrict__ x, double* __restrict__ y) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  double a = 0.0;
  if (row < nrows)
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 16) a += ABS ? fabs(vals[k] * x[col[k]]) : vals[k] * x[col[k]];
  a += __shfl_xor(a, 8);
  a += __shfl_xor(a, 4);
  a += __shfl_xor(a, 2);
  a += __shfl_xor(a, 1);
  if (row < nrows && lane == 0) y[row] = a;
}

// y = A x and ya = |A| |x| in one pass over the matrix (the refinement loop's residual and the scale of its backward error)
static __global__ __launch_bounds__(256) void k_mx_spmv_both(int64_t nrows, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, const double* __restrict__ vals,
                                                             const double* __restrict__ x, double* __restrict__ y,
                                                             double* __restrict__ ya) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  double a = 0.0, b = 0.0;
  if (row < nrows)
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 16) {
      const double t = vals[k] * x[col[k]];
      a += t;
      b += fabs(t);
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if (row < nrows && lane == 0) y[row] = a, ya[row] = b;
}

#define MX_RED 512
// fixed-shape two-stage reductions (bitwise reproducible): partials[b] = sum over the block's slice
static __global__ __launch_bounds__(256) void k_mx_dot(int64_t len, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)MX_RED * 256) s += a[i] * b[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}
static __global__ __launch_bounds__(256) void k_mx_final(int nb, const double* __restrict__ partials, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// y = a*x + b*y   (b == 0: y is not read)
static __global__ void k_mx_axpby(int64_t len, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) y[i] = a * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
}

[[maybe_unused]] static void mx_par_for(int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
  unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 20000) T = 1;
  std::vector<std::thread> th;
  const int64_t chunk = (n + T - 1) / T;
  for (unsigned t = 0; t < T; ++t) {
    const int64_t a = t * chunk, b = std::min<int64_t>(n, a + chunk);
    if (a >= b) break;
    th.emplace_back([=, &fn] { fn(a, b); });
  }
  for (auto& t : th) t.join();
}

// d_out[0] <- rank 0's value on every rank (no-op on a single handle)
static int mx_sync_scalar(MixedBase* h) {
  if (!h->comm || h->comm->size == 1) return PGX_OK;
  if (h->comm->rank != 0) MXHIP(hipMemsetAsync(h->d_out, 0, sizeof(double), h->st));
  const int rc = h->comm->allreduce(h->st, h->d_out, 1);
  if (rc) h->err = "scalar synchronisation: " + h->comm->err;
  return rc;
}

// The replicas of a distributed-LU handle assemble redundantly with atomic-free kernels (pgx_scatter.h), so their vectors are
// bitwise identical.  PGX_CHECK_REPLICAS=1 asserts it (tests): rank 0's copy of v must equal the local one exactly.  Collective.
static int mx_replica_check(MixedBase* h, const double* v, const char* what) {
  if (!h->comm || h->comm->size == 1) return PGX_OK;
  const bool on = [] {
    const char* e = pgx_tune("PGX_CHECK_REPLICAS");
    return e && atoi(e) != 0;
  }();
  if (!on) return PGX_OK;
  MXHIP(hipMemcpyAsync(h->z, v, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));
  if (h->comm->rank != 0) MXHIP(hipMemsetAsync(h->z, 0, sizeof(double) * h->ntot, h->st));
  int rc = h->comm->allreduce(h->st, h->z, (size_t)h->ntot);  // z = rank 0's copy, on every rank
  if (!rc) {
    hipLaunchKernelGGL(k_mx_axpby, dim3((unsigned)((h->ntot + 255) / 256)), dim3(256), 0, h->st, h->ntot, -1.0, v, 1.0, h->z);
    hipLaunchKernelGGL(k_mx_dot, dim3(MX_RED), dim3(256), 0, h->st, h->ntot, h->z, h->z, h->partials);
    hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
    rc = h->comm->allreduce(h->st, h->d_out, 1);  // sum of the ranks' squared differences: every rank sees the verdict
  }
  if (rc) {
    h->err = "replica check: " + h->comm->err;
    return rc;
  }
  MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  if (h->h_out[0] != 0.0) {
    h->err = std::string("replicas of a distributed-LU handle disagree on ") + what;
    return PGX_ECOMM;
  }
  return PGX_OK;
}

static int mx_norm(MixedBase* h, const double* v, double* out, int64_t len = 0) {
  hipLaunchKernelGGL(k_mx_dot, dim3(MX_RED), dim3(256), 0, h->st, len ? len : h->ntot, v, v, h->partials);
  hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
  {
    const int rcs = mx_sync_scalar(h);
    if (rcs) return rcs;
  }
  MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  *out = std::sqrt(h->h_out[0]);
  return PGX_OK;
}

static void mx_axpby(MixedBase* h, double a, const double* x, double b, double* y, int64_t len = 0) {
  const int64_t n = len ? len : h->ntot;
  hipLaunchKernelGGL(k_mx_axpby, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, n, a, x, b, y);
}

static void mx_spmv_dev(MixedBase* h, const double* x, double* y) {
  MxTimer t(h, 4);
  hipLaunchKernelGGL(k_mx_spmv_t<false>, dim3((unsigned)((h->ntot * 16 + 255) / 256)), dim3(256), 0, h->st, h->ntot, h->rowptr,
                     h->col, h->Jv, x, y);
}

// state vectors, reduction scratch, events; the stream must exist
static int mx_alloc_state(MixedBase* h) {
  for (double** v : {&h->x, &h->xk, &h->F, &h->dx, &h->xw, &h->rhs, &h->r, &h->z}) MXALLOC(*v, h->ntot);
  MXALLOC(h->partials, MX_RED);
  MXALLOC(h->d_out, 2);
  MXHIP(hipHostMalloc((void**)&h->h_out, 2 * sizeof(double)));
  MXHIP(hipMemsetAsync(h->x, 0, sizeof(double) * h->ntot, h->st));
  MXHIP(hipMemsetAsync(h->xk, 0, sizeof(double) * h->ntot, h->st));
  hipEventCreate(&h->e0);
  hipEventCreate(&h->e1);
  if (const char* e = pgx_tune("PGX_LAZY_LU")) h->lazy_lu = atoi(e);
  if (const char* e = pgx_tune("PGX_LAZY_BUDGET")) h->lazy_budget = std::max(1, atoi(e));
  if (const char* e = pgx_tune("PGX_MX_REFINE_ETA")) h->refine_eta = atof(e);
  return PGX_OK;
}

static void mx_release(MixedBase* h) {
  hipSetDevice(h->device);
  if (h->st) hipStreamSynchronize(h->st);
  if (pgx_tune("PGX_LAZY_REPORT"))
    fprintf(stderr, "pgx: lazy refactorisation: %ld Newton systems solved with a stale LU (%ld LU solves), %ld attempts fell back\n",
            h->lazy_hits, h->lazy_its, h->lazy_misses);
  if (h->lu) pgx_nd_destroy(h->lu);
  for (void* p : h->allocs) hipFree(p);
  if (h->h_out) hipHostFree(h->h_out);
  if (h->e0) hipEventDestroy(h->e0);
  if (h->e1) hipEventDestroy(h->e1);
  if (h->st) hipStreamDestroy(h->st);
}

static int mx_in(MixedBase* h, double* dst, const double* src, int64_t len = 0) {
  if (!src) return PGX_EINVAL;
  MXHIP(hipMemcpyAsync(dst, src, sizeof(double) * (len ? len : h->ntot), hipMemcpyHostToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
static int mx_out(MixedBase* h, double* dst, const double* src, int64_t len = 0) {
  if (!dst) return PGX_EINVAL;
  MXHIP(hipMemcpyAsync(dst, src, sizeof(double) * (len ? len : h->ntot), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}

static __global__ void k_mx_negate(int64_t len, double* __restrict__ v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) v[i] = -v[i];
}
static __global__ void k_mx_copy_flip(int64_t len, int64_t from, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) y[i] = i >= from ? -x[i] : x[i];
}
// pgx_nd_factor of the current Jacobian / pgx_nd_solve, through the row flip of MixedBase::lu_flip_from
static int mx_lu_factor(MixedBase* h) {
  if (h->lu_flip_from < 0 || h->lu_flip_from >= h->ntot) return pgx_nd_factor(h->lu, h->Jv, 1);
  const int64_t k0 = h->h_rowptr[h->lu_flip_from], len = h->nnz - k0;
  // in place, stream-ordered: negate the tail rows, enqueue the factorisation (every kernel that reads the values is enqueued
  // inside the call), negate back - the exact operator of the refinement keeps its signs
  if (len > 0) hipLaunchKernelGGL(k_mx_negate, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, len, h->Jv + k0);
  const int rc = pgx_nd_factor(h->lu, h->Jv, 1);
  if (len > 0) hipLaunchKernelGGL(k_mx_negate, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, len, h->Jv + k0);
  return rc;
}
static int mx_lu_solve(MixedBase* h, const double* rhs, double* out) {
  if (h->lu_flip_from < 0 || h->lu_flip_from >= h->ntot) return pgx_nd_solve(h->lu, rhs, out, 1);
  if (!h->lu_flip_buf) {
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(double) * (size_t)h->ntot) != hipSuccess) return PGX_ENOMEM;
    h->allocs.push_back(q);
    h->lu_flip_buf = (double*)q;
  }
  hipLaunchKernelGGL(k_mx_copy_flip, dim3((unsigned)((h->ntot + 255) / 256)), dim3(256), 0, h->st, h->ntot, h->lu_flip_from, rhs,
                     h->lu_flip_buf);
  return pgx_nd_solve(h->lu, h->lu_flip_buf, out, 1);
}

// dx = J^{-1} b by LU + iterative refinement on the exact operator; returns the true relative residual
static int mx_dot(MixedBase* h, const double* a, const double* b, double* out);
static int mx_gmres_lu(MixedBase* h, const double* b, double* dx, double bnorm, double tol, int* nsolves, double* relres,
                       int max_cycles = 3, int max_m = 12);
static int mx_linear_solve_ok(MixedBase* h, const double* b, const double* dx, double relres, bool* ok);

// J dx = b by the sparse LU + iterative refinement on the exact operator (the reference: ksp_type preonly + MUMPS).  The LU
// does not pivot across nodes; where refinement alone cannot bring the true relative residual below 1e-7 (late, extremely
// ill-conditioned steps on very fine meshes), the same LU preconditions a short GMRES on the exact operator.
static int mx_linear_solve(MixedBase* h, const double* b, double* dx, const pgx_snes_opts* o, int* nsolves, double* relres) {
  double bnorm = 0, rnorm = 0, prev = 1e300;
  int rc = mx_norm(h, b, &bnorm);
  if (rc) return rc;
  *nsolves = 0;
  if (bnorm == 0.0) {
    MXHIP(hipMemsetAsync(dx, 0, sizeof(double) * h->ntot, h->st));
    *relres = 0.0;
    return PGX_OK;
  }
  const double tol = o->ksp_rtol > 0.0 ? o->ksp_rtol : 1e-10;
  const int maxit = std::max(1, std::min(o->ksp_max_it > 0 ? o->ksp_max_it : 6, 20));
  auto lusolve = [&](const double* rhs, double* out) -> int {
    MxTimer t(h, 3);
    int r2 = mx_lu_solve(h, rhs, out);
    if (r2) h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
    return r2;
  };
  if ((rc = lusolve(b, dx))) return rc;
  ++*nsolves;
  for (int it = 0;; ++it) {
    double anorm = 0;
    {
      MxTimer t(h, 4);
      hipLaunchKernelGGL(k_mx_spmv_both, dim3((unsigned)((h->ntot * 16 + 255) / 256)), dim3(256), 0, h->st, h->ntot, h->rowptr, h->col,
                         h->Jv, dx, h->r, h->z);
    }
    mx_axpby(h, 1.0, b, -1.0, h->r);  // r = b - J dx
    if ((rc = mx_norm(h, h->r, &rnorm))) return rc;
    if ((rc = mx_norm(h, h->z, &anorm))) return rc;
    *relres = rnorm / bnorm;
    // Working precision reached?  The normwise backward error |b - J dx| / (| |J| |dx| | + |b|) is what refinement can drive down; where
    // |J| |dx| dwarfs |b| (late Newton steps: right-hand sides of 1e-8 against |J| |dx| of 1e-1) the RELATIVE residual has a floor of a
    // few 1e-10 that no further solve lowers.  Measured on example 06 at 1024^2: 1.3e-16 ... 1.5e-15 after the first solve, 6.9e-17 after
    // one refinement and from then on - the loop used to spend a third solve on finding that out (round 5).
    const double eta = rnorm / (anorm + bnorm);
    if (o->monitor > 1) printf("      refinement %d  true rel residual %.3e  normwise backward error %.3e\n", it, *relres, eta);
    if (!std::isfinite(*relres) || *relres <= tol || it + 1 >= maxit || *relres > 0.5 * prev || eta <= h->refine_eta) break;
    prev = *relres;
    if ((rc = lusolve(h->r, h->z))) return rc;
    ++*nsolves;
    mx_axpby(h, 1.0, h->z, 1.0, dx);
  }
  const char* ea = pgx_tune("PGX_MX_GMRES_ALWAYS");  // test hook: polish with GMRES whenever refinement stops above tol
  const bool always = ea && atoi(ea);
  if (std::isfinite(*relres) && *relres > 1e-7 && !always) {  // at the rounding level of J itself?  Then GMRES cannot help.
    bool ok = false;
    if ((rc = mx_linear_solve_ok(h, b, dx, *relres, &ok))) return rc;
    if (ok) return PGX_OK;
  }
  if (std::isfinite(*relres) && (*relres > 1e-7 || (always && *relres > tol)))
    return mx_gmres_lu(h, b, dx, bnorm, tol, nsolves, relres);
  return PGX_OK;
}


// One Newton linear system J dx = rhs (Jv holds J at the current iterate).  newton_it = 0: factorise and solve (LU + refinement,
// GMRES safeguard).  Later steps: first the stale factorisation as GMRES preconditioner (see MixedBase::lazy_lu).
static int mx_newton_linear(MixedBase* h, const pgx_snes_opts* opts, int newton_it, int* ns, double* relres) {
  int rc;
  *ns = 0;
  if (h->lazy_lu && newton_it > 0 && h->lu_factored && !h->stale_failed) {
    const double tol = opts->ksp_rtol > 0.0 ? opts->ksp_rtol : 1e-10;
    double bnorm = 0;
    if ((rc = mx_norm(h, h->rhs, &bnorm))) return rc;
    if (bnorm > 0.0 && std::isfinite(bnorm)) {
      MXHIP(hipMemsetAsync(h->dx, 0, sizeof(double) * h->ntot, h->st));
      MXHIP(hipMemcpyAsync(h->r, h->rhs, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));  // r = b - J 0
      *relres = 1.0;
      if ((rc = mx_gmres_lu(h, h->rhs, h->dx, bnorm, tol, ns, relres, 1, h->lazy_budget))) return rc;
      h->lazy_its += *ns;
      if (std::isfinite(*relres) && *relres <= tol) {
        ++h->lazy_hits;
        if (opts->monitor > 1) printf("      stale LU + GMRES: %d solves, true rel residual %.3e\n", *ns, *relres);
        return PGX_OK;
      }
      ++h->lazy_misses;
      h->stale_failed = true;  // the matrix is moving too fast in this solve: factorise from here on
      if (opts->monitor > 1) printf("      stale LU + GMRES gave %.3e after %d solves: refactorising\n", *relres, *ns);
    }
  }
  {
    MxTimer t(h, 2);
    rc = mx_lu_factor(h);
  }
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
    h->lu_factored = false;
    return rc;
  }
  h->lu_factored = true;
  int ns2 = 0;
  rc = mx_linear_solve(h, h->rhs, h->dx, opts, &ns2, relres);
  *ns += ns2;
  return rc;
}

// A linear solve whose true relative residual stays above 1e-7 is a failure (SNES_DIVERGED_LINEAR_SOLVE) - unless the
// residual is at the rounding level of the operator itself: normwise backward error |b - J dx| / (| |J| |dx| | + |b|)
// <= 1e-13.  (Late Newton steps on very fine meshes have right-hand sides of 1e-8 against |J| |dx| of 1e-1: 1e-7 relative
// is then below what fp64 can resolve; the reference's preonly + MUMPS does not look at the residual at all.)
static int mx_linear_solve_ok(MixedBase* h, const double* b, const double* dx, double relres, bool* ok) {
  *ok = std::isfinite(relres) && relres <= 1e-7;
  if (*ok || !std::isfinite(relres)) return PGX_OK;
  double bnorm = 0, anorm = 0;
  int rc = mx_norm(h, b, &bnorm);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mx_spmv_t<true>, dim3((unsigned)((h->ntot * 16 + 255) / 256)), dim3(256), 0, h->st, h->ntot, h->rowptr,
                     h->col, h->Jv, dx, h->z);
  if ((rc = mx_norm(h, h->z, &anorm))) return rc;
  *ok = relres * bnorm <= 1e-13 * (anorm + bnorm);
  return PGX_OK;
}

// SNES newtonls + linesearch none on device `x` (replaced only when reason > 0: lvpp/problem.py:121-123)
static int mx_newton_solve(MixedBase* h, const pgx_snes_opts* opts, int* reason, int* its_out, int* lin_out) {
  if (!opts || !reason) return PGX_EINVAL;
  PgxSolveScope scope(h->st, h->prof, nullptr);
  PgxRange range("pgx:newton_solve");
  const size_t bytes = sizeof(double) * h->ntot;
  int its = 0, lin = 0, rsn = 0, rc = PGX_OK;
  double fnorm = 0, fnorm0 = 0;
  h->stale_failed = false;
  MXHIP(hipMemcpyAsync(h->xw, h->x, bytes, hipMemcpyDeviceToDevice, h->st));
  h->residual_dev(h->xw, h->F);
  if ((rc = mx_replica_check(h, h->F, "the residual"))) return rc;
  if ((rc = mx_norm(h, h->F, &fnorm))) return rc;
  fnorm0 = fnorm;
  if (opts->monitor) printf("  0 SNES Function norm %.12e\n", fnorm);
  if (!std::isfinite(fnorm))
    rsn = PGX_SNES_DIVERGED_FNORM_NAN;
  else if (fnorm < opts->snes_atol)
    rsn = PGX_SNES_CONVERGED_FNORM_ABS;
  const double ttol = fnorm * opts->snes_rtol;
  while (rsn == 0) {
    if (its >= opts->snes_max_it) {
      rsn = PGX_SNES_DIVERGED_MAX_IT;
      break;
    }
    h->jacobian_dev(h->xw);
    mx_axpby(h, -1.0, h->F, 0.0, h->rhs);
    int ns = 0;
    double relres = 0;
    if ((rc = mx_newton_linear(h, opts, its, &ns, &relres))) return rc;
    lin += ns;
    ++its;
    if (opts->monitor) printf("    KSP (LU + %d refinement solves)  true rel residual %.3e\n", ns - 1, relres);
    bool lin_ok = false;
    if ((rc = mx_linear_solve_ok(h, h->rhs, h->dx, relres, &lin_ok))) return rc;
    if (!lin_ok) {
      rsn = PGX_SNES_DIVERGED_LINEAR_SOLVE;
      break;
    }
    mx_axpby(h, 1.0, h->dx, 1.0, h->xw);
    h->residual_dev(h->xw, h->F);
    if ((rc = mx_replica_check(h, h->F, "the residual"))) return rc;
    if ((rc = mx_norm(h, h->F, &fnorm))) return rc;
    if (opts->monitor) printf("  %d SNES Function norm %.12e\n", its, fnorm);
    if (!std::isfinite(fnorm)) {
      rsn = PGX_SNES_DIVERGED_FNORM_NAN;
    } else if (fnorm < opts->snes_atol) {
      rsn = PGX_SNES_CONVERGED_FNORM_ABS;
    } else if (fnorm <= ttol) {
      rsn = PGX_SNES_CONVERGED_FNORM_RELATIVE;
    } else {
      double snorm, xnorm;
      if ((rc = mx_norm(h, h->dx, &snorm))) return rc;
      if ((rc = mx_norm(h, h->xw, &xnorm))) return rc;
      if (snorm < opts->snes_stol * xnorm)
        rsn = PGX_SNES_CONVERGED_SNORM_RELATIVE;
      else if (fnorm > opts->snes_divtol * fnorm0)
        rsn = PGX_SNES_DIVERGED_DTOL;
    }
  }
  if (rsn > 0) MXHIP(hipMemcpyAsync(h->x, h->xw, bytes, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  MXHIP(hipGetLastError());
  if (h->prof) h->ms[5] += scope.stop();
  *reason = rsn;
  if (its_out) *its_out = its;
  if (lin_out) *lin_out = lin;
  return PGX_OK;
}


#define PGX_SNES_DIVERGED_LINE_SEARCH (-6)

// max_i |y_i| / max(|x_i|, 1)  (VecMaxPointwiseDivide of the line search), fixed-shape two-stage reduction
static __global__ __launch_bounds__(256) void k_mx_relmax(int64_t len, const double* __restrict__ y, const double* __restrict__ x,
                                                          double* __restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)MX_RED * 256)
    s = fmax(s, fabs(y[i]) / fmax(fabs(x[i]), 1.0));
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}
static __global__ __launch_bounds__(256) void k_mx_final_max(int nb, const double* __restrict__ partials, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s = fmax(s, partials[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
static int mx_dot(MixedBase* h, const double* a, const double* b, double* out) {
  hipLaunchKernelGGL(k_mx_dot, dim3(MX_RED), dim3(256), 0, h->st, h->ntot, a, b, h->partials);
  hipLaunchKernelGGL(k_mx_final, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
  {
    const int rcs = mx_sync_scalar(h);
    if (rcs) return rcs;
  }
  MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  *out = h->h_out[0];
  return PGX_OK;
}

// Right-preconditioned GMRES(m) on the exact operator with the (inaccurate) LU as preconditioner, from the current dx;
// h->r holds b - J dx on entry.  Modified Gram-Schmidt, Givens rotations, the true residual decides.  At most 3 cycles of 12.
static int mx_gmres_lu(MixedBase* h, const double* b, double* dx, double bnorm, double tol, int* nsolves, double* relres,
                       int max_cycles, int max_m) {
  const int m = 12;
  if (!h->gm_V) {
    const int rca = mx_alloc(h, &h->gm_V, (size_t)(m + 1) * h->ntot);
    if (rca) return rca;
    h->gm_m = m;
  }
  max_m = std::min(std::max(max_m, 1), m);
  auto V = [&](int j) { return h->gm_V + (size_t)j * h->ntot; };
  int rc = PGX_OK;
  for (int cycle = 0; cycle < max_cycles; ++cycle) {
    double beta = 0;
    if ((rc = mx_norm(h, h->r, &beta))) return rc;
    if (!(beta > 0.0) || !std::isfinite(beta)) break;
    mx_axpby(h, 1.0 / beta, h->r, 0.0, V(0));
    std::vector<double> H((m + 1) * m, 0.0), cs(m, 0.0), sn(m, 0.0), g(m + 1, 0.0);
    g[0] = beta;
    int k = 0;
    for (int j = 0; j < max_m; ++j) {
      {
        MxTimer t(h, 3);
        if ((rc = mx_lu_solve(h, V(j), h->z))) {
          h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
          return rc;
        }
      }
      ++*nsolves;
      mx_spmv_dev(h, h->z, V(j + 1));
      for (int i = 0; i <= j; ++i) {
        double hij = 0;
        if ((rc = mx_dot(h, V(j + 1), V(i), &hij))) return rc;
        H[i * m + j] = hij;
        mx_axpby(h, -hij, V(i), 1.0, V(j + 1));
      }
      double hn = 0;
      if ((rc = mx_norm(h, V(j + 1), &hn))) return rc;
      H[(j + 1) * m + j] = hn;
      if (hn > 0.0) mx_axpby(h, 0.0, V(0), 1.0 / hn, V(j + 1));  // scale in place (the x operand is not read: a = 0)
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * H[i * m + j] + sn[i] * H[(i + 1) * m + j];
        H[(i + 1) * m + j] = -sn[i] * H[i * m + j] + cs[i] * H[(i + 1) * m + j];
        H[i * m + j] = t;
      }
      const double a = H[j * m + j], c = H[(j + 1) * m + j], d = std::hypot(a, c);
      cs[j] = d > 0 ? a / d : 1.0;
      sn[j] = d > 0 ? c / d : 0.0;
      H[j * m + j] = d;
      H[(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      k = j + 1;
      if (std::fabs(g[j + 1]) <= tol * bnorm || !(hn > 0.0)) break;
    }
    std::vector<double> y(k, 0.0);
    for (int i = k - 1; i >= 0; --i) {
      double t = g[i];
      for (int l = i + 1; l < k; ++l) t -= H[i * m + l] * y[l];
      y[i] = t / H[i * m + i];
    }
    // dx += M^-1 (V y)
    mx_axpby(h, y[0], V(0), 0.0, h->r);
    for (int i = 1; i < k; ++i) mx_axpby(h, y[i], V(i), 1.0, h->r);
    {
      MxTimer t(h, 3);
      if ((rc = mx_lu_solve(h, h->r, h->z))) {
        h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
        return rc;
      }
    }
    ++*nsolves;
    mx_axpby(h, 1.0, h->z, 1.0, dx);
    mx_spmv_dev(h, dx, h->r);
    mx_axpby(h, 1.0, b, -1.0, h->r);  // r = b - J dx
    double rnorm = 0;
    if ((rc = mx_norm(h, h->r, &rnorm))) return rc;
    *relres = rnorm / bnorm;
    if (!std::isfinite(*relres) || *relres <= tol) break;
  }
  return PGX_OK;
}

// SNES newtonls with the backtracking line search `bt` of order 2 (quadratic): restates PETSc's SNESLineSearchApply_BT
// [upstream, recalled; the same restatement as oracle/qvi_oracle.py::newton_bt] - Armijo parameter 1e-4, maxstep 1e8,
// steptol 1e-12, at most 40 backtracking steps; a non-finite trial residual counts as "no sufficient decrease" and
// shrinks lambda tenfold.  The Jacobian may be a MODIFIED one (J != dF/dx, thermoforming_dolfinx.py:69-71): the initial
// slope uses the same matrix the direction was computed with, as PETSc's MatMult(jac, Y, W) does.
static int mx_newton_solve_bt(MixedBase* h, const pgx_snes_opts* opts, int* reason, int* its_out, int* lin_out) {
  if (!opts || !reason) return PGX_EINVAL;
  PgxSolveScope scope(h->st, h->prof, nullptr);
  PgxRange range("pgx:newton_solve");
  const size_t bytes = sizeof(double) * h->ntot;
  int its = 0, lin = 0, rsn = 0, rc = PGX_OK;
  double fnorm = 0, fnorm0 = 0;
  h->stale_failed = false;
  MXHIP(hipMemcpyAsync(h->xw, h->x, bytes, hipMemcpyDeviceToDevice, h->st));
  h->residual_dev(h->xw, h->F);
  if ((rc = mx_replica_check(h, h->F, "the residual"))) return rc;
  if ((rc = mx_norm(h, h->F, &fnorm))) return rc;
  fnorm0 = fnorm;
  if (opts->monitor) printf("  0 SNES Function norm %.12e\n", fnorm);
  if (!std::isfinite(fnorm))
    rsn = PGX_SNES_DIVERGED_FNORM_NAN;
  else if (fnorm < opts->snes_atol)
    rsn = PGX_SNES_CONVERGED_FNORM_ABS;
  const double ttol = fnorm * opts->snes_rtol;
  while (rsn == 0) {
    if (its >= opts->snes_max_it) {
      rsn = PGX_SNES_DIVERGED_MAX_IT;
      break;
    }
    h->jacobian_dev(h->xw);
    {
      MxTimer t(h, 2);
      rc = mx_lu_factor(h);
    }
    if (rc) {
      h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
      return rc;
    }
    // dx = y = J^{-1} F (PETSc's direction; the update is x - lambda y)
    int ns = 0;
    double relres = 0;
    if ((rc = mx_linear_solve(h, h->F, h->dx, opts, &ns, &relres))) return rc;
    lin += ns;
    ++its;
    if (opts->monitor) printf("    KSP (LU + %d refinement solves)  true rel residual %.3e\n", ns - 1, relres);
    bool lin_ok = false;
    if ((rc = mx_linear_solve_ok(h, h->F, h->dx, relres, &lin_ok))) return rc;
    if (!lin_ok) {
      rsn = PGX_SNES_DIVERGED_LINEAR_SOLVE;
      break;
    }
    double ynorm = 0, initslope = 0, rellength = 0, g = 0;
    if ((rc = mx_norm(h, h->dx, &ynorm))) return rc;
    if (ynorm > 1e8) {
      mx_axpby(h, 0.0, h->dx, 1e8 / ynorm, h->dx);
      ynorm = 1e8;
    }
    mx_spmv_dev(h, h->dx, h->rhs);  // J y
    if ((rc = mx_dot(h, h->F, h->rhs, &initslope))) return rc;
    if (initslope > 0.0) initslope = -initslope;
    if (initslope == 0.0) initslope = -1.0;
    hipLaunchKernelGGL(k_mx_relmax, dim3(MX_RED), dim3(256), 0, h->st, h->ntot, h->dx, h->xw, h->partials);
    hipLaunchKernelGGL(k_mx_final_max, dim3(1), dim3(256), 0, h->st, MX_RED, h->partials, h->d_out);
    if ((rc = mx_sync_scalar(h))) return rc;
    MXHIP(hipMemcpyAsync(h->h_out, h->d_out, sizeof(double), hipMemcpyDeviceToHost, h->st));
    MXHIP(hipStreamSynchronize(h->st));
    rellength = h->h_out[0];
    const double minlambda = 1e-12 / rellength;
    const double f = fnorm * fnorm;
    double lam = 1.0;
    auto trial = [&](double l) -> int {  // z = xw - l y ; r = F(z) ; g = |r|^2
      mx_axpby(h, 1.0, h->xw, 0.0, h->z);
      mx_axpby(h, -l, h->dx, 1.0, h->z);
      h->residual_dev(h->z, h->r);
      double gn = 0;
      int r2 = mx_norm(h, h->r, &gn);
      g = gn * gn;
      return r2;
    };
    auto shrink = [&](double l, bool with_lam) {
      if (!std::isfinite(g)) return 0.1 * l;
      double lt = -initslope / (g - f - 2.0 * (with_lam ? l : 1.0) * initslope);
      lt = std::min(lt, 0.5 * l);
      return lt <= 0.1 * l ? 0.1 * l : lt;
    };
    bool ok = true;
    if ((rc = trial(lam))) return rc;
    if (!(std::isfinite(g) && 0.5 * g <= 0.5 * f + lam * 1e-4 * initslope)) {
      lam = shrink(lam, true);
      if ((rc = trial(lam))) return rc;
      if (!(std::isfinite(g) && 0.5 * g < 0.5 * f + lam * 1e-4 * initslope)) {
        int count = 0;
        while (true) {
          if (lam <= minlambda) {
            ok = false;
            break;
          }
          lam = shrink(lam, false);
          if ((rc = trial(lam))) return rc;
          if (std::isfinite(g) && 0.5 * g < 0.5 * f + lam * 1e-4 * initslope) break;
          if (++count > 40) {
            ok = false;
            break;
          }
        }
      }
    }
    if (opts->monitor > 1) printf("      line search: lambda %.6e  gnorm %.12e\n", lam, std::sqrt(g));
    if (!ok) {
      rsn = PGX_SNES_DIVERGED_LINE_SEARCH;
      break;
    }
    MXHIP(hipMemcpyAsync(h->xw, h->z, bytes, hipMemcpyDeviceToDevice, h->st));
    MXHIP(hipMemcpyAsync(h->F, h->r, bytes, hipMemcpyDeviceToDevice, h->st));
    fnorm = std::sqrt(g);
    if (opts->monitor) printf("  %d SNES Function norm %.12e\n", its, fnorm);
    if (fnorm < opts->snes_atol) {
      rsn = PGX_SNES_CONVERGED_FNORM_ABS;
    } else if (fnorm <= ttol) {
      rsn = PGX_SNES_CONVERGED_FNORM_RELATIVE;
    } else {
      double xnorm;
      if ((rc = mx_norm(h, h->xw, &xnorm))) return rc;
      if (lam * ynorm < opts->snes_stol * xnorm)
        rsn = PGX_SNES_CONVERGED_SNORM_RELATIVE;
      else if (fnorm > opts->snes_divtol * fnorm0)
        rsn = PGX_SNES_DIVERGED_DTOL;
    }
  }
  if (rsn > 0) MXHIP(hipMemcpyAsync(h->x, h->xw, bytes, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  MXHIP(hipGetLastError());
  if (h->prof) h->ms[5] += scope.stop();
  *reason = rsn;
  if (its_out) *its_out = its;
  if (lin_out) *lin_out = lin;
  return PGX_OK;
}

// SNES newtonls with the `l2` line search (examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py:66-79:
// snes_linesearch_type l2, maxlambda 1): restates PETSc's SNESLineSearchApply_L2 [upstream, recalled; the same restatement as
// oracle/ic_oracle.py::newton_l2] - |F|^2 sampled at lambda_old = 0, the midpoint and lambda = 1, ONE secant step on its
// derivative (PETSc's default max_it of this search), the update kept only inside [steptol, maxlambda] = [1e-12, 1]; a non-finite
// end-point residual halves lambda.  The update is x - lambda y with y = J^{-1} F.
[[maybe_unused]] static int mx_newton_solve_l2(MixedBase* h, const pgx_snes_opts* opts, int* reason, int* its_out, int* lin_out) {
  if (!opts || !reason) return PGX_EINVAL;
  PgxSolveScope scope(h->st, h->prof, nullptr);
  PgxRange range("pgx:newton_solve");
  const size_t bytes = sizeof(double) * h->ntot;
  const double steptol = 1e-12, maxlambda0 = 1.0;
  int its = 0, lin = 0, rsn = 0, rc = PGX_OK;
  double fnorm = 0, fnorm0 = 0;
  h->stale_failed = false;
  MXHIP(hipMemcpyAsync(h->xw, h->x, bytes, hipMemcpyDeviceToDevice, h->st));
  h->residual_dev(h->xw, h->F);
  if ((rc = mx_norm(h, h->F, &fnorm))) return rc;
  fnorm0 = fnorm;
  if (opts->monitor) printf("  0 SNES Function norm %.12e\n", fnorm);
  if (!std::isfinite(fnorm))
    rsn = PGX_SNES_DIVERGED_FNORM_NAN;
  else if (fnorm < opts->snes_atol)
    rsn = PGX_SNES_CONVERGED_FNORM_ABS;
  const double ttol = fnorm * opts->snes_rtol;
  auto trial = [&](double l, double* g) -> int {  // z = xw - l y ; r = F(z) ; g = |r|^2
    mx_axpby(h, 1.0, h->xw, 0.0, h->z);
    mx_axpby(h, -l, h->dx, 1.0, h->z);
    h->residual_dev(h->z, h->r);
    double gn = 0;
    const int r2 = mx_norm(h, h->r, &gn);
    *g = gn * gn;
    return r2;
  };
  while (rsn == 0) {
    if (its >= opts->snes_max_it) {
      rsn = PGX_SNES_DIVERGED_MAX_IT;
      break;
    }
    h->jacobian_dev(h->xw);
    {
      MxTimer t(h, 2);
      rc = mx_lu_factor(h);
    }
    if (rc) {
      h->err = std::string("direct solver: ") + pgx_nd_last_error(h->lu);
      return rc;
    }
    int ns = 0;
    double relres = 0;
    if ((rc = mx_linear_solve(h, h->F, h->dx, opts, &ns, &relres))) return rc;
    lin += ns;
    ++its;
    if (opts->monitor) printf("    KSP (LU + %d refinement solves)  true rel residual %.3e\n", ns - 1, relres);
    bool lin_ok = false;
    if ((rc = mx_linear_solve_ok(h, h->F, h->dx, relres, &lin_ok))) return rc;
    if (!lin_ok) {
      rsn = PGX_SNES_DIVERGED_LINEAR_SOLVE;
      break;
    }
    double lam = 1.0, lam_old = 0.0, maxl = maxlambda0, fn_old = fnorm * fnorm, fm = 0, fe = 0;
    double lam_mid = 0.5 * (lam + lam_old);
    bool failed = false;
    for (int i = 0; i < 1; ++i) {  // -snes_linesearch_max_it of l2: 1
      while (true) {
        if ((rc = trial(lam_mid, &fm))) return rc;
        if ((rc = trial(lam, &fe))) return rc;
        if (std::isfinite(fe)) break;
        if (lam <= steptol) {
          failed = true;
          break;
        }
        maxl = 0.95 * lam;
        lam = 0.5 * (lam + lam_old);
        lam_mid = 0.5 * (lam + lam_old);
      }
      if (failed) break;
      const double dl = lam - lam_old;
      const double d1 = (3.0 * fe - 4.0 * fm + fn_old) / dl, d1_old = (-3.0 * fn_old + 4.0 * fm - fe) / dl;
      const double d2 = (d1 - d1_old) / dl;
      double upd;
      if (d2 > 0.0)
        upd = lam - d1 / d2;
      else if (d2 < 0.0)
        upd = lam + d1 / d2;
      else
        break;
      if (upd < steptol) upd = 0.5 * (lam + lam_old);
      if (!std::isfinite(upd) || upd > maxl) break;
      lam_old = lam, lam = upd, fn_old = fe;
      lam_mid = 0.5 * (lam + lam_old);
    }
    if (failed) {
      rsn = PGX_SNES_DIVERGED_LINE_SEARCH;
      break;
    }
    double g = 0;
    if ((rc = trial(lam, &g))) return rc;
    MXHIP(hipMemcpyAsync(h->xw, h->z, bytes, hipMemcpyDeviceToDevice, h->st));
    MXHIP(hipMemcpyAsync(h->F, h->r, bytes, hipMemcpyDeviceToDevice, h->st));
    fnorm = std::sqrt(g);
    if (opts->monitor > 1) printf("      line search: lambda %.6e\n", lam);
    if (opts->monitor) printf("  %d SNES Function norm %.12e\n", its, fnorm);
    if (!std::isfinite(fnorm)) {
      rsn = PGX_SNES_DIVERGED_FNORM_NAN;
    } else if (fnorm < opts->snes_atol) {
      rsn = PGX_SNES_CONVERGED_FNORM_ABS;
    } else if (fnorm <= ttol) {
      rsn = PGX_SNES_CONVERGED_FNORM_RELATIVE;
    } else {
      double xnorm, ynorm;
      if ((rc = mx_norm(h, h->dx, &ynorm))) return rc;
      if ((rc = mx_norm(h, h->xw, &xnorm))) return rc;
      if (ynorm < opts->snes_stol * xnorm)
        rsn = PGX_SNES_CONVERGED_SNORM_RELATIVE;
      else if (fnorm > opts->snes_divtol * fnorm0)
        rsn = PGX_SNES_DIVERGED_DTOL;
    }
  }
  if (rsn > 0) MXHIP(hipMemcpyAsync(h->x, h->xw, bytes, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  MXHIP(hipGetLastError());
  if (h->prof) h->ms[5] += scope.stop();
  *reason = rsn;
  if (its_out) *its_out = its;
  if (lin_out) *lin_out = lin;
  return PGX_OK;
}

static int mx_profile(MixedBase* h, int enable, double ms[6]) {
  if (ms)
    for (int i = 0; i < 6; ++i) ms[i] = h->ms[i];
  for (int i = 0; i < 6; ++i) h->ms[i] = 0;
  h->prof = enable != 0;
  return PGX_OK;
}
