/*
 * pgx_nd.h - C ABI of the sparse direct solver inside libpgx.so: geometric nested-dissection multifrontal LU on the GPU.
 *
 * Replaces, for the Newton linear systems of this repository's hot path, what the reference requests from PETSc with
 *     "ksp_type": "preonly", "pc_type": "lu", "pc_factor_mat_solver_type": "mumps"
 * (examples/01_obstacle_problem/obstacle_pg.py:129-131; examples/06_gradient_constraints/
 *  gradient_constraint_dolfinx.py:118-121 incl. "mat_mumps_icntl_14"; examples/02_signorini/signorini_dolfinx.py
 *  petsc_options), i.e. MatLUFactorSymbolic / MatLUFactorNumeric / MatSolve of a MUMPS-backed PC:
 *
 *   pgx_nd_create   PCSetUp symbolic phase (ordering + symbolic factorisation), once per sparsity pattern
 *   pgx_nd_factor   MatLUFactorNumeric, once per Newton step
 *   pgx_nd_solve    MatSolve / KSPSolve(preonly)
 *
 * The matrix is a general CSR matrix with structurally symmetric pattern; dofs are grouped into NODES (all dofs living
 * on one mesh entity: u, psi, ... of a vertex or edge midpoint).  Nodes are eliminated as blocks in a nested-dissection
 * order obtained by recursive coordinate bisection; no pivoting across nodes (the saddle-point Newton matrices of the
 * LVPP examples are strongly factorisable in any node order, see oracle/nd_proto.py and DESIGN.md).
 * Plain pointers and sizes only.  Returns 0 or a negative PGX_E* code (include/pgx.h); text via pgx_nd_last_error.
 */
#ifndef PGX_ND_H
#define PGX_ND_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_nd pgx_nd;

typedef struct {
  int64_t n;                 /* matrix order */
  const int32_t* rowptr;     /* [n+1] host; column indices sorted within each row */
  const int32_t* col;        /* [nnz] host */
  int32_t n_nodes;
  const int32_t* node_of_dof;/* [n] host: node owning each dof */
  int32_t dim;               /* 2 or 3 */
  const double* node_coords; /* [n_nodes][dim] host */
  int32_t leaf_nodes;        /* stop bisecting below this many nodes; 0 = default (16) */
} pgx_nd_matrix;

/* Host-only symbolic statistics (no GPU needed). */
typedef struct {
  int64_t n_fronts, n_levels, max_front;   /* max padded front order */
  int64_t arena_doubles;                   /* device storage: compact padded factors + the two working buffers */
  int64_t factor_nnz;                      /* unpadded entries kept (L and U) */
  double flops;                            /* unpadded factorisation flops */
  double flops_padded;                     /* flops the level-batched kernels execute */
  int64_t perturbed_pivots;                /* (near-)zero pivots the LAST completed factorisation replaced by +-1e-300 (static
                                            * perturbation, no pivoting): > 0 means its solves are unreliable; pgx_nd_last_error
                                            * then says so.  pgx_nd_get_stats waits for a factorisation in flight. */
} pgx_nd_stats;

/* device < 0: symbolic phase only (no GPU touched) - for pgx_nd_get_stats / pgx_nd_export_* on CPU-only machines. */
/* Environment: PGX_ND_CUT_GB (default 160) - a factorisation whose device storage exceeds this many GB cuts the tree at depth 3
 * and factorises the subtrees below one after the other, so that the working buffers only hold one subtree's fronts
 * (0: always, negative: never). */
int pgx_nd_create(const pgx_nd_matrix* A, int device, void* hip_stream /* may be NULL: own stream */, pgx_nd** out);
/* Distributed factorisation over the ranks of a pgx_comm (include/pgx.h: pgx_comm_rccl_init / pgx_comm_local_group; size a
 * power of two): the dissection tree is cut at depth log2(size); every rank factorises one subtree in its own HBM, rank 0
 * also the levels above, receiving the subtree roots' Schur blocks (one grouped ncclSend/ncclRecv per factorisation, one
 * small pair per solve, one all-reduce of the solution).  Every rank passes the SAME matrix (pattern at create, values at
 * factor, right-hand side at solve) and receives the full solution; all calls are collective.  Replaces what MUMPS does
 * over MPI in the reference (`mpirun -n N` + "pc_factor_mat_solver_type": "mumps"). */
struct pgx_comm;
int pgx_nd_create_dist(const pgx_nd_matrix* A, struct pgx_comm* comm, int device, void* hip_stream, pgx_nd** out);
void pgx_nd_destroy(pgx_nd* s);
const char* pgx_nd_last_error(const pgx_nd* s);
int pgx_nd_get_stats(const pgx_nd* s, pgx_nd_stats* st);

/* Numeric factorisation of the matrix whose CSR values (same pattern as at create) are DEVICE-resident (on_device != 0)
 * or on the host.  Asynchronous on the solver's stream when on_device; pgx_nd_solve orders after it. */
int pgx_nd_factor(pgx_nd* s, const double* vals, int on_device);
/* x = A^{-1} b; b, x device pointers (on_device != 0) or host; b == x allowed. */
int pgx_nd_solve(pgx_nd* s, const double* b, double* x, int on_device);
/* SYMMETRIC matrices (round 5): the Newton matrices of examples 01 and 06 are symmetric indefinite saddle-point matrices, that of
 * example 02 becomes one when the rows of the latent block are negated.  After pgx_nd_set_symmetric(s, 1) pgx_nd_factor reads the
 * caller's values as those of a symmetric matrix (both triangles still passed, the pattern is the general one) and computes
 * L D L^T in LU clothing: half the flops - the U panels are written as scaled transposes of the L panels, the Schur updates run on
 * the tiles on and below the diagonal.  The factors it stores and pgx_nd_solve are those of the general path (same accuracy class:
 * no pivoting across blocks either way).  The request is honoured with the default kernels (parent-centric assembly, MFMA panels) - single
 * rank or distributed, cut schedule or not - and silently ignored otherwise (pgx_nd_is_symmetric tells); a matrix that is NOT symmetric gets the factorisation of its lower triangle's
 * symmetric completion - wrap the solve in refinement on the exact operator, as every caller in this library does.
 * Replaces: -pc_factor_mat_solver_type mumps with -mat_mumps_sym / MatSetOption(A, MAT_SYMMETRIC, PETSC_TRUE). */
int pgx_nd_set_symmetric(pgx_nd* s, int on);
int pgx_nd_is_symmetric(const pgx_nd* s);
/* Test hooks, host arithmetic only (callable without a GPU): the enumeration of the tiles a symmetric GEMM launch computes - the tiles
 * (tr, tc) of an nr x nc rectangle with tc <= tr + band, in groups of 8 tile rows walked column by column. */
int pgx_nd_sym_tile_count(int nr, int nc, int band);
void pgx_nd_sym_tile_at(int t, int nr, int nc, int band, int* tr, int* tc);
/* accumulated device time of the last factor / solve calls in ms (HIP events; 0 until pgx_nd_timing(s,1)) */
int pgx_nd_timing(pgx_nd* s, int enable, double* factor_ms, double* solve_ms);

/* Diagnostic: device time per TREE DEPTH (0 = root) of the factorisations, forward sweeps and backward sweeps since recording was
 * switched on (HIP events on the solver's stream at the depth boundaries).  enable > 0: start recording and clear the sums,
 * 0: stop and clear, < 0: leave the state alone (read only).  *n_depths: in = capacity of the arrays, out = number of tree
 * depths; any array may be NULL; calls[3] = factorisations / forward / backward sweeps summed.  The arrays are filled BEFORE
 * the state changes, so one call can read and stop. */
int pgx_nd_depth_profile(pgx_nd* s, int enable, int32_t* n_depths, double* factor_ms, double* fwd_ms, double* bwd_ms,
                         int32_t* calls);

/* Symbolic structure, for tests (the numpy emulation in tests/test_nd_symbolic.py factorises with exactly these maps).
 * Call with NULL arrays to get sizes.  Layout: fronts are numbered batch by batch ("slots"); a batch = the fronts of one
 * tree depth and one size class, batches ordered by depth (root first);
 * batch l holds slots [lev_start[l], lev_start[l+1]) and pads every front to pivot order P[l], border B[l], M = P+B,
 * stored column-major at offset lev_off[l] + (slot - lev_start[l]) * M*M of a VIRTUAL arena in which every front is a
 * full M x M matrix (the device keeps the factors compactly and the M x M matrices only while a tree depth is being
 * factorised - same local positions, other base addresses).  Local index of an own dof k: k; of the
 * s-th border dof: P + s.  rel[rel_ptr[f] + s] = local index in the PARENT's front of border dof s of front f. */
int pgx_nd_export_levels(const pgx_nd* s, int64_t* n_levels, int64_t* lev_start, int32_t* P, int32_t* B, int64_t* lev_off,
                         int32_t* depth /* tree depth of each batch; the children of depth d live at depth d+1 */);
int pgx_nd_export_fronts(const pgx_nd* s, int64_t* n_fronts, int32_t* fp, int32_t* fb, int32_t* parent, int32_t* slot01,
                         int64_t* dof_ptr, int32_t* own_dofs, int64_t* rel_ptr, int32_t* rel);
int pgx_nd_export_dest(const pgx_nd* s, int64_t* nnz, int64_t* dest);
/* Tests: the symbolic structure rank `rank` of `size` would build in pgx_nd_create_dist (no GPU, no communicator), and its
 * distribution data: tree depth of the cut, batch holding the subtree roots, this rank's root slot, and on rank 0 the slots
 * of all subtree roots (own + ghosts; dest[] entries of fronts living on other ranks are -1). */
int pgx_nd_create_symbolic_dist(const pgx_nd_matrix* A, int rank, int size, pgx_nd** out);
int pgx_nd_export_dist(const pgx_nd* s, int32_t* kdist, int32_t* kbatch, int32_t* root_slot, int32_t* ghost_slot /* [size] */);

#ifdef __cplusplus
}
#endif
#endif
