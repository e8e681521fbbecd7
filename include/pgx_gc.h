/*
 * pgx_gc.h - C ABI of libpgx.so for the LVPP Newton inner loop of example 06 (gradient constraint |grad u| <= phi,
 * vector latent variable): everything below `problem.solve()` in
 * examples/06_gradient_constraints/gradient_constraint_dolfinx.py:179, i.e. what the reference delegates to DOLFINx
 * assembly + PETSc SNES (newtonls, linesearch none) + MUMPS LU (:108-131).
 *
 * Spaces (reference :38-46 with primal_degree = 2): u in P2 (dofs = [vertices | one per edge], as pgx_mesh.cell_dofs),
 * psi in (P1)^2.  State / residual vectors have length n_dofs + 2 * n_vertices:
 *        x = [u_0 .. u_{n2-1} | psi_x(0 .. nv-1) | psi_y(0 .. nv-1)]
 * Residual (:100-107, degree-10 measure :53), w0 = previous proximal iterate:
 *        R_u   = alpha (grad u, grad v) + (psi - psi0, grad v) - alpha (f, v)
 *        R_psi = (grad u, w) - (phi psi / sqrt(1 + |psi|^2), w)
 * Jacobian = derivative (NonlinearProblem default): [[alpha K, G^T],[G, -N(psi)]].
 *
 *   pgx_gc_create         NonlinearProblem(F, u=sol, bcs, petsc_options) construction (:108-131)
 *   pgx_gc_set/get_state  sol.x.array access (:97-98,194)
 *   pgx_gc_set/get_prev, pgx_gc_advance_prev   w0.x.array[:] = sol.x.array (:205)
 *   pgx_gc_set_alpha      alpha.value = ... (:172-177)
 *   pgx_gc_residual / pgx_gc_jacobian_fill / pgx_gc_csr_export / pgx_gc_spmv   the SNES callbacks and the PETSc Mat
 *                         (lvpp twin: src/lvpp/problem.py:54-77,110)
 *   pgx_gc_newton_solve   problem.solve() (:179) with SNES reason / iteration count (:180-183)
 *   pgx_gc_l2_increment   assemble_scalar(dot(diff, diff) dx) + allreduce + sqrt (:164-166,184-186)
 * Conventions as in pgx.h: 0 or a negative PGX_E* code; caller owns host buffers; handle owns device memory; synchronous.
 * The Newton linear systems are solved by the sparse direct solver of pgx_nd.h (the reference's pc_type lu) with
 * iterative refinement on the exact operator.  There is no CPU fallback.
 */
#ifndef PGX_GC_H
#define PGX_GC_H
#include <stdint.h>

#include "pgx.h"
#include "pgx_nd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_gc_handle pgx_gc_handle;

typedef struct {
  int32_t nq;             /* quadrature points per cell (<= 40); degree 10 in the reference (:53) */
  const double* qpts;     /* [nq][2] reference triangle */
  const double* qwts;     /* [nq], sum 1/2 */
  const double* phi_dofs; /* [n_dofs] nodal values of phi interpolated into the P2 primal space (:55-56) */
  const double* f_dofs;   /* [n_dofs] nodal values of f (:60-61) */
  int32_t n_bc;           /* Dirichlet dofs of u (:63-69,109-111) */
  const int32_t* bc_dofs;
  const double* bc_vals;  /* NULL = homogeneous */
} pgx_gc_problem;

/* General primal degree k = 2..8 (gradient_constraint_dolfinx.py:245-250; latent degree k - 1): the caller states both Lagrange
 * spaces - dof maps and the basis at the quadrature points - and the library runs the same algorithm with table-driven element
 * kernels.  State layout x = [u (n_u) | psi_x (n_p) | psi_y (n_p)]; phi_dofs / f_dofs / bc_dofs of pgx_gc_problem refer to the
 * n_u primal dofs.  (proximalgalerkin_amd/lagrange.py builds these tables for the equispaced Lagrange elements; a DOLFINx binding
 * would pass V.sub(0).collapse() / V.sub(1).collapse() dofmaps and basix tabulations.) */
typedef struct {
  int32_t n_vertices, n_cells;
  const double* coords;         /* [n_vertices][2] */
  const int32_t* cells;         /* [n_cells][3] vertex ids spanning the AFFINE cell map x0 + (x1-x0) xi + (x2-x0) eta: a triangle's
                                   vertices, or origin / +xi corner / +eta corner of a parallelogram (--cell_type quadrilateral,
                                   :229-236: create_unit_square's rectangles; the rule of pgx_gc_problem is then one on the unit
                                   square, weights summing to 1) */
  int32_t nu, np;               /* local nodes of the primal / latent element: triangles (k+1)(k+2)/2 <= 45, k(k+1)/2 <= 36;
                                   quadrilaterals (k+1)^2 <= 81, k^2 <= 64 */
  int32_t n_u, n_p;             /* global dofs of the primal space / of ONE latent component */
  const int32_t* cell_dofs_u;   /* [n_cells][nu] */
  const int32_t* cell_dofs_p;   /* [n_cells][np] */
  const double* coords_u;       /* [n_u][2] node coordinates (nested-dissection ordering only) */
  const double* coords_p;       /* [n_p][2] */
  const double* tab_Nu;         /* [nq][nu] primal basis at the quadrature points of pgx_gc_problem */
  const double* tab_dNu;        /* [nq][nu][2] its REFERENCE gradients */
  const double* tab_Np;         /* [nq][np] latent basis */
} pgx_gc_spaces;
int pgx_gc_create_general(const pgx_gc_spaces* spaces, const pgx_gc_problem* prob, int device, pgx_gc_handle** out);

/* mesh: pgx_mesh with cell_dofs / n_dofs set (P2); structured_nx/ny are ignored */
int pgx_gc_create(const pgx_mesh* mesh, const pgx_gc_problem* prob, int device, pgx_gc_handle** out);
/* one handle per GPU over a pgx_comm: replicated iterate and assembly, distributed sparse LU (see pgx_sg_create_dist) */
int pgx_gc_create_dist(const pgx_mesh* mesh, const pgx_gc_problem* prob, pgx_comm* comm, int device, pgx_gc_handle** out);
/* symbolic statistics of the handle's sparse LU (flop counts, arena size: include/pgx_nd.h) */
int pgx_gc_lu_stats(const pgx_gc_handle* h, pgx_nd_stats* st);
/* 1: the handle's sparse LU runs in symmetric mode (pgx_nd_set_symmetric: the Newton matrix of example 06 is symmetric) - the
 * factorisation then executes about half of pgx_nd_stats.flops */
int pgx_gc_lu_is_symmetric(const pgx_gc_handle* h);
void pgx_gc_destroy(pgx_gc_handle* h);
const char* pgx_gc_last_error(const pgx_gc_handle* h);
int pgx_gc_num_dofs(const pgx_gc_handle* h, int64_t* ntot);
int pgx_gc_set_state(pgx_gc_handle* h, const double* x);
int pgx_gc_get_state(pgx_gc_handle* h, double* x);
int pgx_gc_set_prev(pgx_gc_handle* h, const double* x);
int pgx_gc_get_prev(pgx_gc_handle* h, double* x);
int pgx_gc_advance_prev(pgx_gc_handle* h);
int pgx_gc_set_alpha(pgx_gc_handle* h, double alpha);
/* x == NULL: device state.  F may be NULL. */
int pgx_gc_residual(pgx_gc_handle* h, const double* x, double* F, double* fnorm);
int pgx_gc_jacobian_fill(pgx_gc_handle* h, const double* x);
/* mixed CSR matrix of the last fill (BC rows/cols of u = identity); NULL arrays -> sizes only */
int pgx_gc_csr_export(pgx_gc_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* vals);
int pgx_gc_spmv(pgx_gc_handle* h, const double* x, double* y);
/* opts: snes_* and ksp_rtol (true relative residual of the refined LU solve; 0 = 1e-10, the target of example 01's Newton solves; 1e-12 until round 5), ksp_max_it (refinement steps) */
int pgx_gc_newton_solve(pgx_gc_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its);
int pgx_gc_l2_increment(pgx_gc_handle* h, double* out); /* || u - u_prev ||_L2 */
/* accumulated device ms since the last reset: [0] residual [1] jacobian [2] LU factor [3] LU solves [4] spmv [5] total */
int pgx_gc_profile(pgx_gc_handle* h, int enable, double ms[6]);

#ifdef __cplusplus
}
#endif
#endif
