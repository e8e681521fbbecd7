/*
 * pgx_sg.h - C ABI of libpgx.so for the LVPP Newton inner loop of example 02 (Signorini contact of a 3-D linear-elastic
 * body with a rigid plane; latent variable on the contact surface): everything below `solver.solve()` in
 * examples/02_signorini/signorini_dolfinx.py:333, i.e. what the reference delegates to DOLFINx assembly (cell + exterior
 * facet integrals on a submesh, :199-249) + PETSc SNES (newtonls, linesearch none) + MUMPS LU (:271-291).
 *
 * Spaces at degree 1 (BASELINE.json config 5): u in (P1)^3 on a tetrahedral mesh, psi in P1 on the contact facets; degree 2 (the
 * reference's default, :68-73; pgx_sg_mesh.degree = 2): u in (P2)^3, psi in P2 on the contact facets, affine cells; hexahedra
 * (the reference's native mesh, :376-383; pgx_sg_mesh.cell_type = 1) with Q1 / Q2 elements on parallelepipeds.
 *        x = [u_x (nv) | u_y (nv) | u_z (nv) | psi (one per contact node, ordered by node id)],  nv = vertices (degree 1) / P2 nodes
 * Residual (:236-249), n_g = -e_z, g = x_z - gap, f = 0:
 *        R_u   = alpha (sigma(u), eps(v)) - <psi - psi_k, v.n_g>_Gamma
 *        R_psi = <u.n_g, w>_Gamma + <exp(psi), w>_Gamma - <g, w>_Gamma
 * Jacobian [[alpha A, +M_G],[-M_G, D(psi)]] (nonsymmetric sign pattern, as UFL's derivative gives it).
 *
 *   pgx_sg_create         NonlinearProblem(F, [u, psi], bcs, entity_maps, petsc_options) construction (:281-291)
 *   pgx_sg_set/get_state, pgx_sg_set/get_prev, pgx_sg_advance_prev   u.x.array / psi.x.array / psi_k, u_prev (:336-343)
 *   pgx_sg_set_alpha      alpha.value = ... (:323-328)
 *   pgx_sg_residual / pgx_sg_jacobian_fill / pgx_sg_csr_export / pgx_sg_spmv   SNES callbacks and the PETSc Mat
 *   pgx_sg_newton_solve   solver.solve() (:333) with reason / iteration count (:334-335); tolerances as set by
 *                         solver.solver.setTolerances(atol=, rtol=) (:331-332)
 *   pgx_sg_u_increment    ||u - u_prev||_2 (vector 2-norm, :337-339)
 * Conventions as in pgx.h.  Linear solves: sparse LU of pgx_nd.h + iterative refinement.  No CPU fallback.
 */
#ifndef PGX_SG_H
#define PGX_SG_H
#include <stdint.h>

#include "pgx.h"
#include "pgx_nd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_sg_handle pgx_sg_handle;

typedef struct {
  int32_t n_vertices, n_cells;
  const double* coords;    /* [n_vertices][3] */
  const int32_t* cells;    /* [n_cells][4] */
  int32_t n_facets;        /* potential-contact facets (facet_tag.find(contact), :186-189) */
  const int32_t* facets;   /* [n_facets][3] vertex ids */
  int32_t degree;          /* 0 or 1: degree 1 as described above.  2 (the reference's default, :68-73): u in (P2)^3, psi in P2 on the
                            * contact facets.  Then n_vertices counts the P2 NODES (mesh vertices, then one node per edge; `coords`
                            * holds the edge midpoints for them), cells is [n_cells][10]: 4 vertices, then the edge nodes of
                            * (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); facets is [n_facets][6]: 3 vertices, then the edge nodes of
                            * (0,1) (0,2) (1,2).  Cells are affine (geometry from the vertices) unless created by pgx_sg_create_curved.  bc_dofs: component * n_nodes +
                            * node; pgx_sg_contact_vertices returns the node of every psi dof. */
  int32_t cell_type;       /* 0: tetrahedra (above).  1: hexahedra - the reference's native mesh (:376-383) - with Q_d elements, d = degree
                            * (0 / 1 -> Q1, 2 -> Q2): cells is [n_cells][(d+1)^3], facets [n_facets][(d+1)^2] (quadrilaterals), local nodes
                            * numbered lexicographically (xi fastest), basis = tensor Lagrange on the equispaced nodes k/d; cells and facets
                            * must be affine images of the cube / square (parallelepipeds: true for box grids).  The facet quadrature of
                            * pgx_sg_problem is then a rule on the unit SQUARE (weights sum to 1). */
} pgx_sg_mesh;

typedef struct {
  double E, nu, gap;       /* :58-66 */
  int32_t nq;              /* facet quadrature (degree 4 in the reference, :67-69), <= 16 points */
  const double* qpts;      /* [nq][2] reference triangle */
  const double* qwts;      /* [nq], sum 1/2 */
  int32_t n_bc;            /* Dirichlet dofs of u: component * n_vertices + vertex (:255-268) */
  const int32_t* bc_dofs;
  const double* bc_vals;   /* NULL = homogeneous */
} pgx_sg_problem;

int pgx_sg_create(const pgx_sg_mesh* mesh, const pgx_sg_problem* prob, int device, pgx_sg_handle** out);
/* ORDER-2 GEOMETRY (round 5): the reference's half sphere is a mesh of 10-node tetrahedra (src/lvpp/mesh_generation.py:88,158
 * `order=2`) and DOLFINx integrates on the curved cells.  With pgx_sg_mesh.degree = 2 the ten nodes of a cell ARE its geometry nodes
 * (`coords` of an edge node = the mesh's mid-edge node instead of the midpoint): the discretisation is isoparametric P2; with degree 1
 * the mesh arrays hold the vertices only (P1 fields on the quadratic cells).  In both cases the geometry itself enters through two
 * tables a binding reads off the coordinate element: */
typedef struct {
  int32_t nq;              /* cell quadrature on the reference tetrahedron (weights sum 1/6).  The reference leaves this integral's
                            * degree to UFL's estimator (signorini_dolfinx.py:237-239 has no metadata on dx); <= 512 points */
  const double* qpts;      /* [nq][3] */
  const double* qwts;      /* [nq] */
  const double* cell_geo;  /* [n_cells][nq][10]: |det J|, then J^-1 row-major (d xi_k / d x_d) of x(xi) = sum_a X_a N2_a(xi) */
  const double* facet_geo; /* [n_facets][prob->nq][2]: surface element |x_xi x x_eta| of the 6-node contact facet and its z
                            * coordinate at the facet quadrature points of pgx_sg_problem */
} pgx_sg_curved;
/* Single handle, tetrahedra, degree 1 or 2 (PGX_EINVAL otherwise).  Everything else as pgx_sg_create. */
int pgx_sg_create_curved(const pgx_sg_mesh* mesh, const pgx_sg_problem* prob, const pgx_sg_curved* curved, int device,
                         pgx_sg_handle** out);
/* One handle per GPU over a pgx_comm (BASELINE.json config 5: 4 GPUs).  The ELEMENTS are partitioned (round 4): the cells are cut
 * into comm->size slabs of equal count along the longest axis of the mesh, every rank assembles the elasticity blocks of its own
 * slab only and one all-reduce sums the constant matrix - the owned-cell assembly of a distributed DOLFINx mesh
 * (signorini_dolfinx.py:283-291, `kind="mpi"`); the sparse LU, 94 % of a Newton step, is distributed (pgx_nd_create_dist).  Still
 * replicated: the mesh arrays handed in, the iterate, and what a Newton step assembles on top of the constant matrix (a product
 * with it and the contact facets: < 1 % of a step).  Replaces `mpirun -n N python signorini_dolfinx.py`; all calls are collective
 * and return identical results on every rank.  Tuning key PGX_SG_PARTITION=0: every rank assembles every cell (rounds 2-3). */
int pgx_sg_create_dist(const pgx_sg_mesh* mesh, const pgx_sg_problem* prob, pgx_comm* comm, int device, pgx_sg_handle** out);
/* cells whose element matrices THIS rank assembled / cells of the mesh (equal on a single handle) */
int pgx_sg_partition_info(const pgx_sg_handle* h, int64_t* owned_cells, int64_t* total_cells);
/* symbolic statistics of the handle's sparse LU (flop counts, arena size: include/pgx_nd.h) */
int pgx_sg_lu_stats(const pgx_sg_handle* h, pgx_nd_stats* st);
/* 1: the handle's sparse LU runs in symmetric mode (the latent rows are negated on the way into it: csrc/pgx_mixed.h lu_flip_from) */
int pgx_sg_lu_is_symmetric(const pgx_sg_handle* h);
void pgx_sg_destroy(pgx_sg_handle* h);
const char* pgx_sg_last_error(const pgx_sg_handle* h);
int pgx_sg_num_dofs(const pgx_sg_handle* h, int64_t* ntot, int64_t* npsi);
int pgx_sg_contact_vertices(const pgx_sg_handle* h, int32_t* verts /* [npsi]: vertex of each psi dof */);
int pgx_sg_set_state(pgx_sg_handle* h, const double* x);
int pgx_sg_get_state(pgx_sg_handle* h, double* x);
int pgx_sg_set_prev(pgx_sg_handle* h, const double* x);
int pgx_sg_get_prev(pgx_sg_handle* h, double* x);
int pgx_sg_advance_prev(pgx_sg_handle* h);
int pgx_sg_set_alpha(pgx_sg_handle* h, double alpha);
int pgx_sg_residual(pgx_sg_handle* h, const double* x, double* F, double* fnorm);
int pgx_sg_jacobian_fill(pgx_sg_handle* h, const double* x);
int pgx_sg_csr_export(pgx_sg_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col, double* vals);
int pgx_sg_spmv(pgx_sg_handle* h, const double* x, double* y);
int pgx_sg_newton_solve(pgx_sg_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its);
int pgx_sg_u_increment(pgx_sg_handle* h, double* out);
int pgx_sg_profile(pgx_sg_handle* h, int enable, double ms[6]);

#ifdef __cplusplus
}
#endif
#endif
