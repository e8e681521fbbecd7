/*
 * pgx_ic.h - C ABI of libpgx.so for the LVPP Newton inner loop of example 08 (intersecting constraints: an obstacle u >= phi0
 * AND a gradient bound |u'| <= phi on one primal field, two latent variables; SURVEY.md section 8(f) rank 3 names its residual as
 * the front end's acceptance test): everything below `problem.solve()` in
 * examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py:127, i.e. DOLFINx assembly + PETSc SNES newtonls with
 * the `l2` line search + MUMPS LU (:66-79).
 *
 * Mixed [P1, P1, (P1)^1] on an interval mesh (:13-23); x = [u (nv) | psi0 (nv) | psi (nv)].  Residual (:47-58):
 *     R_u    = alpha [(u', v') + (c, v)] + (psi0 - psi0_iter, v) + (psi - psi_iter, v')
 *     R_psi0 = (u, w0) - (exp(psi0), w0) - (phi0, w0)                       example 01's latent row
 *     R_psi  = (u', w) - (phi psi / sqrt(1 + psi^2), w)                     example 06's latent row
 * Jacobian = the exact derivative (:75-77).  phi0 and phi are UFL expressions of the coordinate in the reference (:36-45); here
 * the host samples them at the quadrature points (like `phi_q` of pgx.h), and pgx_ic_set_phi replaces the gradient bound when the
 * script changes `phic.value` (:116).
 *
 *   pgx_ic_create          NonlinearProblem(F, z, bcs=bcs, petsc_options=sp) construction (:75-77,124-126)
 *   pgx_ic_set/get_state, set/get_prev, advance_prev    z.x.array, z_iter.interpolate(z) (:119,171), z.interpolate(z_iter) (:147)
 *   pgx_ic_set_alpha       alpha.value = 1, /= 2, *= r (:118,144,166-168)
 *   pgx_ic_set_phi         phic.value = phi_ (:116)
 *   pgx_ic_residual / pgx_ic_jacobian_fill / pgx_ic_csr_export / pgx_ic_spmv    SNES callbacks and the PETSc Mat
 *   pgx_ic_newton_solve    problem.solve() (:127): opts->linesearch = 2 selects the l2 line search
 *   pgx_ic_l2_increment    sqrt(assemble_scalar(L2_u)) = ||u - u_iter||_L2 (:81,156)
 * Conventions as in pgx.h.  Linear solves: sparse LU of pgx_nd.h (the chain of vertices dissected like any other graph) +
 * iterative refinement.  No CPU fallback.
 */
#ifndef PGX_IC_H
#define PGX_IC_H
#include <stdint.h>

#include "pgx.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_ic_handle pgx_ic_handle;

typedef struct {
  int32_t n_vertices;     /* nv >= 2; cell e joins vertices e and e + 1 (create_unit_interval's topology, :13) */
  const double* x;        /* [nv] vertex coordinates, strictly increasing */
  int32_t nq;             /* quadrature points per cell (<= 16): one rule for every integral */
  const double* qpts;     /* [nq] in (0, 1) */
  const double* qwts;     /* [nq], sum 1 */
  const double* phi0_q;   /* [nv - 1][nq] obstacle at the physical quadrature points (:39-42) */
  const double* phi_q;    /* [nv - 1][nq] gradient bound (:44-45) */
  double c;               /* the Constant of E (:30-32) */
  int32_t n_bc;           /* Dirichlet vertices of u (homogeneous, :60-63) */
  const int32_t* bc_dofs;
} pgx_ic_problem;

int pgx_ic_create(const pgx_ic_problem* prob, int device, pgx_ic_handle** out);
void pgx_ic_destroy(pgx_ic_handle* h);
const char* pgx_ic_last_error(const pgx_ic_handle* h);
int pgx_ic_num_dofs(const pgx_ic_handle* h, int64_t* ntot);
int pgx_ic_set_state(pgx_ic_handle* h, const double* x);
int pgx_ic_get_state(pgx_ic_handle* h, double* x);
int pgx_ic_set_prev(pgx_ic_handle* h, const double* x);
int pgx_ic_get_prev(pgx_ic_handle* h, double* x);
int pgx_ic_advance_prev(pgx_ic_handle* h);
int pgx_ic_set_alpha(pgx_ic_handle* h, double alpha);
int pgx_ic_set_phi(pgx_ic_handle* h, const double* phi_q); /* [nv - 1][nq] */
int pgx_ic_residual(pgx_ic_handle* h, const double* x, double* F, double* fnorm);
int pgx_ic_jacobian_fill(pgx_ic_handle* h, const double* x);
int pgx_ic_csr_export(pgx_ic_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* vals);
int pgx_ic_spmv(pgx_ic_handle* h, const double* x, double* y);
int pgx_ic_newton_solve(pgx_ic_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its);
int pgx_ic_l2_increment(pgx_ic_handle* h, double* out);
int pgx_ic_profile(pgx_ic_handle* h, int enable, double ms[6]);

#ifdef __cplusplus
}
#endif
#endif
