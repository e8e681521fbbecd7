/*
 * pgx_qvi.h - C ABI of libpgx.so for the LVPP Newton inner loop of example 05 (thermoforming quasi-variational inequality:
 * membrane u, mould temperature T, latent variable psi): everything below
 * `problem.solve()` in examples/05_obstacle_type_qvi/thermoforming_dolfinx.py:124, i.e. DOLFINx assembly + PETSc SNES
 * newtonls with the `bt` line search of order 2 + MUMPS LU (:100-113), with a MODIFIED Jacobian J != dF/ds (:69-71).
 *
 * Mixed [P1, P1, P1] on a triangulation (:28-33); x = [u (nv) | T (nv) | psi (nv)].  Residual (:62-67):
 *     R_u   = alpha (grad u, grad v) + (psi - psi_prev, v) - alpha (f, v)
 *     R_T   = (grad T, grad q) + beta (T, q) - (g(exp(-psi)), q)           g piecewise linear with knee q0 (:36-48)
 *     R_psi = (u, w) + (exp(-psi), w) - (Phi0 + xi T, w)                    Phi0, xi functions of the coordinates (:58-59)
 * Jacobian = derivative(F - eps/alpha (grad psi, grad w), s) (:69-71).
 *
 *   pgx_qvi_create        NonlinearProblem(F, u=s, bcs=[bc], J=J, petsc_options=sp) construction (:114-116)
 *   pgx_qvi_set/get_state, set/get_prev, advance_prev   s.x.array, s_prev.x.array[:] = s.x.array (:119,156)
 *   pgx_qvi_set_alpha     alpha.value *= 4 ... (:157-158)
 *   pgx_qvi_residual / pgx_qvi_jacobian_fill / pgx_qvi_csr_export / pgx_qvi_spmv   SNES callbacks and the PETSc Mat
 *   pgx_qvi_newton_solve  problem.solve() (:124): opts->linesearch = 1 selects the bt line search, 0 the full step
 *   pgx_qvi_h1_increment  sqrt(assemble_scalar(u_diff_H1)) (:81-83,139-140)
 * Conventions as in pgx.h.  Linear solves: sparse LU of pgx_nd.h + iterative refinement.  No CPU fallback.
 */
#ifndef PGX_QVI_H
#define PGX_QVI_H
#include <stdint.h>

#include "pgx.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_qvi_handle pgx_qvi_handle;

typedef struct {
  int32_t nq;             /* quadrature points per cell (<= 16): one rule for every integral */
  const double* qpts;     /* [nq][2] */
  const double* qwts;     /* [nq], sum 1/2 */
  double beta, f, knee, eps_mod; /* :55-57 beta = 1, f = 25; :36 knee = 0.01; :70 eps = 1e-10 */
  int32_t n_bc;           /* Dirichlet vertices of u (homogeneous, :73-79) */
  const int32_t* bc_dofs;
} pgx_qvi_problem;

/* mesh: pgx_mesh with n_vertices, n_cells, coords, cells (cell_dofs / structured_* ignored) */
int pgx_qvi_create(const pgx_mesh* mesh, const pgx_qvi_problem* prob, int device, pgx_qvi_handle** out);
void pgx_qvi_destroy(pgx_qvi_handle* h);
const char* pgx_qvi_last_error(const pgx_qvi_handle* h);
int pgx_qvi_num_dofs(const pgx_qvi_handle* h, int64_t* ntot);
int pgx_qvi_set_state(pgx_qvi_handle* h, const double* x);
int pgx_qvi_get_state(pgx_qvi_handle* h, double* x);
int pgx_qvi_set_prev(pgx_qvi_handle* h, const double* x);
int pgx_qvi_get_prev(pgx_qvi_handle* h, double* x);
int pgx_qvi_advance_prev(pgx_qvi_handle* h);
int pgx_qvi_set_alpha(pgx_qvi_handle* h, double alpha);
int pgx_qvi_residual(pgx_qvi_handle* h, const double* x, double* F, double* fnorm);
int pgx_qvi_jacobian_fill(pgx_qvi_handle* h, const double* x);
int pgx_qvi_csr_export(pgx_qvi_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* vals);
int pgx_qvi_spmv(pgx_qvi_handle* h, const double* x, double* y);
int pgx_qvi_newton_solve(pgx_qvi_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its);
int pgx_qvi_h1_increment(pgx_qvi_handle* h, double* out);
int pgx_qvi_profile(pgx_qvi_handle* h, int enable, double ms[6]);

#ifdef __cplusplus
}
#endif
#endif
