/*
 * pgx.h - C ABI of libpgx.so: MI355X-native LVPP / proximal-Galerkin Newton inner loop.
 *
 * Drop-in boundary for ONE hot path of METHODS-Group/ProximalGalerkin: everything below
 * `problem.solve()` in examples/01_obstacle_problem/obstacle_pg.py:190, i.e. what the reference
 * delegates to DOLFINx assembly + PETSc SNES/KSP + MUMPS.  Plain pointers and sizes only; no torch
 * or C++ types cross this boundary.  All reals are IEEE fp64, all indices int32.
 *
 * DOF layout of every state / residual vector of length 2*n_vertices:
 *      x = [u_0 .. u_{n-1}, psi_0 .. psi_{n-1}]        (n = n_vertices for P1, pgx_mesh.n_dofs for P2)
 *
 * Conventions:
 *   - every function returns 0 on success, a negative PGX_E* code otherwise; pgx_last_error() gives text.
 *   - the caller owns all host buffers passed in; they are never retained past the call.
 *   - the handle owns all device memory; a handle is bound to one HIP device and is not thread-safe.
 *   - calls are synchronous: the handle's stream is idle when a call returns.
 *   - convergence is reported with PETSc's signed SNESConvergedReason values (>0 converged).
 *
 * Reference interface each entry point replaces (paths relative to the reference checkout):
 *   pgx_create            dolfinx.fem.petsc.NonlinearProblem(F,u,bcs,J,petsc_options) construction
 *                         (examples/01_obstacle_problem/obstacle_pg.py:140-142): form compilation,
 *                         sparsity pattern, SNES creation; lvpp twin SNESProblem.__init__ +
 *                         SNESSolver.create_data_structures (src/lvpp/problem.py:14-52,106-112)
 *   pgx_set_state/get_state   sol.x.array[:] access (obstacle_pg.py:157,226)
 *   pgx_set_prev / pgx_advance_prev   sol_k.x.array[:] = sol.x.array[:] (obstacle_pg.py:158,226)
 *   pgx_zero_state        sol.x.array[:] = 0.0; sol_k.x.array[:] = sol.x.array[:] (obstacle_pg.py:157-158)
 *   pgx_set_alpha         alpha.value = ... (obstacle_pg.py:175-186)
 *   pgx_residual          SNESProblem.F(snes, x, F) (src/lvpp/problem.py:54-67)
 *   pgx_jacobian_fill     SNESProblem.J(snes, x, J, P) (src/lvpp/problem.py:69-77)
 *   pgx_csr_export        the PETSc Mat `A` of SNESSolver (src/lvpp/problem.py:110), for inspection
 *   pgx_spmv              MatMult on that Mat (inside KSP; the reference uses LU instead:
 *                         obstacle_pg.py:129-131)
 *   pgx_newton_solve      problem.solve() / SNESSolver.solve() (obstacle_pg.py:190;
 *                         src/lvpp/problem.py:114-124) incl. "copy back only if converged"
 *   pgx_observables       the six assemble_scalar + allreduce calls (obstacle_pg.py:145-152,196-201)
 *   pgx_destroy           SNESSolver.__del__ (src/lvpp/problem.py:126-127)
 */
#ifndef PGX_H
#define PGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_OK 0
#define PGX_EINVAL (-1)   /* bad argument / inconsistent sizes */
#define PGX_EHIP (-2)     /* HIP runtime error (text in pgx_last_error) */
#define PGX_ENODEV (-3)   /* no usable GPU */
#define PGX_ENOMEM (-4)
#define PGX_ESTATE (-5)   /* call order violated (e.g. spmv before jacobian_fill) */
#define PGX_ECOMM (-6)    /* communicator failure / a peer rank did not arrive (text in pgx_last_error) */

/* PETSc SNESConvergedReason values mirrored by pgx_newton_solve */
#define PGX_SNES_CONVERGED_FNORM_ABS 2
#define PGX_SNES_CONVERGED_FNORM_RELATIVE 3
#define PGX_SNES_CONVERGED_SNORM_RELATIVE 4
#define PGX_SNES_DIVERGED_LINEAR_SOLVE (-3)
#define PGX_SNES_DIVERGED_FNORM_NAN (-4)
#define PGX_SNES_DIVERGED_MAX_IT (-5)
#define PGX_SNES_DIVERGED_DTOL (-9)

typedef struct pgx_handle pgx_handle;

typedef struct {
  int32_t n_vertices;
  int32_t n_cells;
  const double* coords;   /* [n_vertices][2] */
  const int32_t* cells;   /* [n_cells][3], vertex ids */
  /* >0 only for the right-diagonal structured triangulation with vertex v = j*(nx+1)+i and cells
   * (2q, 2q+1) = [v0,v1,v3],[v0,v2,v3] of square q = j*nx+i (dolfinx create_rectangle default).
   * Enables the geometric multigrid preconditioner; 0 = general mesh (single-level smoother). */
  int32_t structured_nx;
  int32_t structured_ny;
  /* degree 2 only (NULL / 0 for degree 1): dofs per field are [vertices | one per edge]; row c lists the 3
   * vertex ids of cell c (== cells[c]) then its 3 edge dofs (>= n_vertices), local edge i OPPOSITE local
   * vertex i (the Basix reference-triangle convention the reference's dofmap uses). */
  const int32_t* cell_dofs; /* [n_cells][6] */
  int32_t n_dofs;           /* n_vertices + n_edges */
} pgx_mesh;

typedef struct {
  int32_t degree;          /* Lagrange degree of both fields: 1 or 2 (obstacle_pg.py -p) */
  int32_t nq;              /* quadrature points per cell (<= 16) */
  const double* qpts;      /* [nq][2] on the reference triangle */
  const double* qwts;      /* [nq], summing to 1/2 */
  const double* phi_q;     /* [n_cells][nq] obstacle at physical quadrature points (obstacle_pg.py:107-111) */
  double f;                /* constant forcing (obstacle_pg.py:74) */
  int32_t n_bc;            /* Dirichlet dofs of the u block (obstacle_pg.py:76-83) */
  const int32_t* bc_dofs;  /* [n_bc] vertex ids */
  const double* bc_vals;   /* [n_bc] or NULL for homogeneous */
} pgx_problem;

/* Subset of the PETSc option dictionary the reference passes (obstacle_pg.py:128-139). */
typedef struct {
  double snes_rtol;   /* default 1e-8; ex 01 sets 1e-6 */
  double snes_atol;   /* 1e-50 */
  double snes_stol;   /* 1e-8 */
  double snes_divtol; /* 1e4 */
  int32_t snes_max_it;/* 50; ex 01 sets 100 */
  /* Newton linear solve (replaces ksp preonly + pc lu/mumps): FGMRES + multigrid V-cycle */
  double ksp_rtol;    /* relative TRUE residual target; 0 (default) = auto: 1e-10 for P1, 1e-11 for P2, chosen from the
                         measured effect on the final primal field */
  int32_t ksp_max_it; /* default 200; a sharded P2 handle (no sparse-LU fallback) raises the DEFAULT to 400, an explicit value is kept */
  int32_t ksp_restart;/* default 30 (basis storage allows up to 50) */
  int32_t mg_nu;      /* pre/post smoothing sweeps, default 6 (even values run as fused double sweeps) */
  double mg_omega;    /* collective-Jacobi damping, default 0.75 */
  int32_t monitor;    /* 1 = print per-Newton-step residuals (snes_monitor/ksp_monitor) */
  int32_t pc_type;    /* preconditioner of the FGMRES Newton solve: 0 = auto (1 for P1 on a structured mesh, 2 for P2 and for general
                         meshes), 1 = multigrid V-cycle (single-level smoother on general meshes),
                         2 = sparse LU (nested-dissection multifrontal, include/pgx_nd.h) - what the reference's
                         "pc_type": "lu" (obstacle_pg.py:130) asks for; FGMRES then acts as iterative refinement */
  int32_t linesearch; /* snes_linesearch_type: 0 = none / basic (full step: examples 01, 02, 06), 1 = bt of order 2
                         (quadratic backtracking, examples/05_obstacle_type_qvi/thermoforming_dolfinx.py:104,111); honoured by
                         the handles of pgx_qvi.h, pgx_gc.h and pgx_sg.h (pgx_newton_solve of example 01 takes the full step) */
} pgx_snes_opts;

void pgx_default_opts(pgx_snes_opts* o);

int pgx_create(const pgx_mesh* mesh, const pgx_problem* prob, int device, pgx_handle** out);
/* The same for a mesh with ORDER-2 GEOMETRY (6-node triangles: the reference's own meshes are gmsh meshes of element order 2,
 * examples/01_obstacle_problem/generate_mesh_gmsh.py:30-33, and DOLFINx then integrates on the curved cells).  geoq
 * [n_cells][nq][5] holds, for every quadrature point of prob->qpts on every cell, |det J| and the four entries of J^-1 (row-major:
 * d xi_k / d x_d) of the quadratic cell map x(xi) = sum_a X_a N2_a(xi) - what a binding reads off the coordinate element
 * (fem.Mesh.geometry_at here); prob->phi_q is phi at the PHYSICAL quadrature points of that map.  Unstructured, unpartitioned
 * meshes; degree 2 = isoparametric P2 (csrc/pgx_p2.hip), degree 1 = hat functions on the quadratic cells, the reference's default
 * run on its own meshes (the k_*_c kernels of csrc/pgx_kernels.hip): residual, Jacobian blocks, b_phi and the observables evaluate
 * weights and physical gradients per quadrature point, nothing is constant per cell.  A structured or partitioned mesh is
 * PGX_EINVAL (those have affine cells by construction). */
int pgx_create_curved(const pgx_mesh* mesh, const pgx_problem* prob, const double* geoq, int device, pgx_handle** out);
void pgx_destroy(pgx_handle* h);
const char* pgx_last_error(const pgx_handle* h); /* h may be NULL: error of the last failed pgx_create */

int pgx_num_dofs(const pgx_handle* h, int64_t* ndofs);          /* 2 * dofs per field */
int pgx_set_state(pgx_handle* h, const double* x);               /* host -> device `sol` */
int pgx_get_state(pgx_handle* h, double* x);
int pgx_set_prev(pgx_handle* h, const double* xk);               /* host -> device `sol_k` */
int pgx_get_prev(pgx_handle* h, double* xk);
int pgx_advance_prev(pgx_handle* h);                             /* sol_k <- sol on device */
int pgx_zero_state(pgx_handle* h);                               /* sol = sol_k = 0 on device (obstacle_pg.py:157-158) */
int pgx_set_alpha(pgx_handle* h, double alpha);

/* F(x) with the callback contract of problem.py:54-67. x==NULL -> use device `sol`. F may be NULL. */
int pgx_residual(pgx_handle* h, const double* x, double* F, double* fnorm);
/* Fill the Jacobian values at x (x==NULL -> device `sol`) into the fixed pattern. */
int pgx_jacobian_fill(pgx_handle* h, const double* x);

/* Scalar pattern shared by the four blocks [[alpha*K, M],[M, -D]]:
 * call once with arrays NULL to get sizes, then with host buffers to receive copies.
 * K, M are the unconstrained blocks; bc rows/cols are applied by the operator. */
int pgx_csr_export(pgx_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col,
                   double* K, double* M, double* D);

/* y = J x with J = Jacobian of the last pgx_jacobian_fill, incl. BC rows/cols (identity). */
int pgx_spmv(pgx_handle* h, const double* x, double* y);
/* Time `reps` SpMV launches on device-resident data with HIP events on the handle's stream, in the cache state of a solve
 * rather than back to back (the matrix-free kernel's footprint would fit the 256 MB Infinity Cache): multigrid handles replay
 * "one V-cycle producing z, then J z" as an FGMRES iteration does and subtract the time of the V-cycles alone; other handles
 * sweep 512 MB of idle storage between two applies.  Returns the average ms per launch and the algorithmic bytes one launch
 * moves. */
int pgx_spmv_bench(pgx_handle* h, int reps, double* avg_ms, double* algorithmic_bytes);
/* The same measurement with every apply preceded by a 512 MB sweep of unrelated device storage: the kernel's operands come from
 * HBM, not from the 256 MB Infinity Cache (the operator of the 2048^2 benchmark, 273 MB, would otherwise mostly sit in it).
 * bench.py reports it as roofline.cold_frac next to the in-solve figure. */
int pgx_spmv_bench_cold(pgx_handle* h, int reps, double* avg_ms, double* algorithmic_bytes);
/* Sharded handles: collectives issued by THIS rank since the last reset - [0] halo exchanges (one grouped send/recv batch with
 * both strip neighbours each), [1] all-reduces, [2] V-cycles, [3] Krylov iterations.  Zeros on unsharded handles. */
int pgx_comm_counts(pgx_handle* h, int64_t out[4], int reset);
/* Operator apply of the outer Krylov solver ("SpMV").  kind 1 (default on structured P1 meshes): matrix-free stencil kernel
 * k_st_spmv_r - K, M are the uniform mesh's constants, D(psi) its half-stored 7-point stencil, 65 B per vertex; kind 0: the
 * block-CSR stream kernel k_bspmv_stream (what general meshes and P2 always use, 232 B per P1 row); kind 2: generic
 * one-thread-per-vertex stencil kernel (A/B); kind -1: query only.  *active (may be NULL) receives the kind that will run on
 * this handle (0 when the mesh has no grid structure).  pgx_spmv / pgx_spmv_bench follow the selection; pgx_spmv_bench's
 * `bytes` are the algorithmic bytes of the selected kernel.  P2 handles (round 4): kind 3 (or 1; the default on the uniform
 * structured mesh) = the structured apply of csrc/pgx_p2st.hip - interior vertex / edge groups through a 46-entry table, no column
 * indices, D(psi) from a structure-of-arrays copy, 496 B per group; the frame rows in CSR form -, kind 0 = the block-CSR kernel
 * k_bspmv_bal; *active is 3 or 0. */
int pgx_spmv_select(pgx_handle* h, int kind, int* active);
/* P2 handles: out = {state, i0, ni, j0, nj} of the structured apply - state 0: not available on this mesh (general meshes,
 * fewer than 8 cells per side, a pattern or K / M entries that are not those of the uniform right-diagonal mesh), 1: pattern verified,
 * 2: in use; the interior groups are the vertices (i, j), i0 <= i < i0 + ni, j0 <= j < j0 + nj, with their three edges. */
int pgx_p2_stencil_info(pgx_handle* h, int32_t out[5]);
/* (P2 handles: pgx_smoother_bench times one additive patch sweep - k_patch_apply + k_patch_edges - on the inverses of the last
 * Jacobian; `bytes` = inverses + dof table + residual and iterate of the patch dofs + the parked edge contributions.) */
/* The same for the TIME-DOMINANT kernel of the multigrid-preconditioned solve: the fused three-sweep smoother of the finest
 * level (k_st_smoothR, post-smoothing variant: x + P x_c folded in), at the Jacobian of the last pgx_jacobian_fill.
 * algorithmic_bytes = one pass over the level: 4 D-stencil arrays + b (2) + x (2) + the coarse correction (2 arrays of n/4) read,
 * the new iterate (2) written.  PGX_ESTATE on meshes without a grid hierarchy. */
int pgx_smoother_bench(pgx_handle* h, int reps, double* avg_ms, double* algorithmic_bytes);

/* Tuning / A-B / test switches (kernel variants, tile sizes, thresholds, consistency checks such as PGX_CHECK_REPLICAS).
 * libpgx.so never reads them from the environment: they exist only in a process-wide table filled through this call
 * (value NULL clears the key; keys start with "PGX_").  Most are read when a handle is created.  The environment variables the
 * library itself honours are the three documented run-time options: PGX_COMM_TIMEOUT (seconds a transport wait may take),
 * PGX_ROCTX (roctx ranges around the solver phases) and PGX_ND_THREADS (host threads of the symbolic factorisation).
 * tools/ and tests/ opt in to the old behaviour through the Python loader (PGX_TUNING_FROM_ENV=1 copies PGX_* variables into the
 * table before each create). */
int pgx_tuning_set(const char* key, const char* value);
/* Measurement aid: average device time (HIP events on the handle's stream, launches back to back) of the part of one V(nu,nu)
 * cycle that starts on multigrid level `level` (0 = the whole preconditioner application; -1 = the fused tail launch only) with the
 * default nu and damping.  *n_level receives the vertex count of that level.  bench.py reports the "coarse part" (levels of at most
 * 513^2 vertices) with it.  Same preconditions as pgx_smoother_bench. */
int pgx_vcycle_bench(pgx_handle* h, int level, int reps, double* avg_ms, int* n_level);

/* One nonlinear solve from device `sol` with proximal centre `sol_k`; on reason>0 `sol` is replaced. */
int pgx_newton_solve(pgx_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its);
/* energy, |complementarity|, feasibility, dual feasibility, H1 increment, latent L2 increment */
int pgx_observables(pgx_handle* h, double out[6]);

/* Accumulated device-time per phase since the last reset, ms (HIP events; only when enabled):
 * [0] residual [1] jacobian fill [2] mg setup [3] spmv (outer Krylov) [4] v-cycle [5] orthogonalisation
 * [6] observables [7] total newton_solve wall */
int pgx_profile_enable(pgx_handle* h, int on);
int pgx_profile_get(pgx_handle* h, double ms[8], int reset);

/* ---------------------------------------------------------------------------------------------------------
 * Sharded path: one handle per GPU, the structured mesh cut into horizontal strips of
 * vertex rows.  Replaces what the reference inherits from DOLFINx/PETSc over MPI:
 *   Vec.ghostUpdate(INSERT, FORWARD)   src/lvpp/problem.py:56,58,71,73  -> halo exchange of ghost vertex rows
 *   F.ghostUpdate(ADD, REVERSE)        src/lvpp/problem.py:66           -> not needed: ghost cells are assembled
 *                                                                          redundantly (owner computes whole rows)
 *   comm.allreduce(scalar)             obstacle_pg.py:50 (six calls :196-201), SNES/KSP norms and dots
 *                                                                       -> ONE packed all-reduce per use
 * Every rank passes its LOCAL strip (owned rows + ghost rows, see pgx_partition_rows) as an ordinary structured
 * pgx_mesh / pgx_problem; vectors crossing the ABI are local (owned + ghost entries), like DOLFINx's x.array.
 * All calls on a sharded handle are COLLECTIVE: every rank of the communicator must make the same calls in the
 * same order.  Iteration counts and results equal the single-handle solve (same algebra; only the summation
 * order of dot products differs).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pgx_comm pgx_comm; /* opaque: neighbour halo exchange + all-reduce(sum) on device buffers */

typedef struct pgx_partition {
  int32_t rank, size;   /* strip index (0 = lowest rows) and number of strips */
  int32_t global_ny;    /* cell rows of the GLOBAL mesh; must be divisible by size * 2^dist_levels */
  int32_t dist_levels;  /* multigrid levels kept distributed (ghost depth 2^dist_levels on the fine grid);
                           coarser levels are replicated on every rank.  0 = choose (at most 3) */
} pgx_partition;

/* Which global vertex rows the local mesh of `part->rank` must contain: rows [row0, row0+nrows), of which
 * [own0, own0+nown) are owned (the others are ghosts).  Fills part->dist_levels when it was 0.  No GPU needed. */
int pgx_partition_rows(pgx_partition* part, int32_t* row0, int32_t* nrows, int32_t* own0, int32_t* nown);

/* RCCL transport, one process per GPU (torch.distributed launch): rank 0 creates the id, the host broadcasts
 * the 128 bytes by any means, every rank calls init.  ncclSend/ncclRecv to the two strip neighbours over xGMI,
 * ncclAllReduce for packed scalars - all enqueued on the handle's stream (no host synchronisation). */
int pgx_comm_rccl_unique_id(char id[128]);
int pgx_comm_rccl_init(const char id[128], int rank, int size, int device, pgx_comm** out);
/* In-process transport: `size` communicators for `size` handles driven by `size` host threads of ONE process
 * (any mix of devices, also all on one GPU).  Same call sequence as RCCL through host-synchronised device
 * copies; used by the test-suite to run the sharded algorithm on a single-GPU box. */
int pgx_comm_local_group(int size, pgx_comm** out /* [size] */);
/* Inter-PROCESS transport through a POSIX shared-memory segment `name` ("/..."; rank 0 creates it, the others attach, the name
 * is unlinked once all ranks are in), host-staged: one communicator per process, any mix of devices - also several processes on
 * ONE GPU, which RCCL refuses.  A torch.distributed.run launch with this transport executes everything the multi-GPU launch
 * does except RCCL's byte movement (mpirun-style process model of the reference: obstacle_pg.py:64, problem.py:56-73).
 * slot_bytes: mailbox per rank (0 = 64 MiB; larger payloads move in chunks).  host_mode != 0: the buffers handed to the
 * operations are HOST memory (protocol tests on machines without a GPU).  A peer that does not arrive within
 * PGX_COMM_TIMEOUT seconds (default 120) fails every rank with PGX_ECOMM instead of hanging. */
int pgx_comm_shm_init(const char* name, int rank, int size, uint64_t slot_bytes, int host_mode, pgx_comm** out);
/* the four operations of a communicator, callable directly (tests): in-place sum over the ranks; exchange of the entries
 * [send_lo, +n_send_lo) / [send_hi, +n_send_hi) of f0 (and f1 if non-null) with rank-1 / rank+1 into [recv_lo, ...) /
 * [recv_hi, ...); rank r > 0 -> rank 0 at recv0 + r*n; rank 0's send0 + r*n -> rank r.  Buffers: device memory, or host
 * memory for a host_mode communicator. */
int pgx_comm_allreduce(pgx_comm* c, double* buf, uint64_t n);
int pgx_comm_halo(pgx_comm* c, double* f0, double* f1, uint64_t send_lo, uint64_t n_send_lo, uint64_t recv_lo, uint64_t n_recv_lo,
                  uint64_t send_hi, uint64_t n_send_hi, uint64_t recv_hi, uint64_t n_recv_hi);
int pgx_comm_gather0(pgx_comm* c, const double* send, uint64_t n, double* recv0);
int pgx_comm_scatter0(pgx_comm* c, const double* send0, uint64_t n, double* recv);
/* Self-check before the first solve: one halo exchange with both strip neighbours and one packed all-reduce on a known pattern,
 * verified, each bounded by timeout_s (<= 0: 10 s).  Collective.  host != 0 for a host_mode communicator.  On failure the text
 * (pgx_comm_last_error) names the operation that failed or did not complete; the caller should end the process (a collective
 * that hangs cannot be cancelled).  Replaces nothing in the reference - MPI_Init fails loudly by itself; RCCL over a mis-wired
 * launch does not. */
int pgx_comm_selfcheck(pgx_comm* c, int host, double timeout_s);
void pgx_comm_free(pgx_comm* c);
const char* pgx_comm_last_error(void);

int pgx_create_sharded(const pgx_mesh* local_mesh, const pgx_problem* local_prob, const pgx_partition* part,
                       pgx_comm* comm, int device, pgx_handle** out);
/* Replicated handles with a DISTRIBUTED sparse LU (BASELINE.json config 3: P2 on 8 GPUs, where the multigrid path is not
 * robust and one GPU cannot hold the factor): every rank passes the WHOLE mesh / problem and keeps the whole iterate;
 * assembly, SpMV and the Krylov vectors are replicated, the LU preconditioner - the dominant cost - is factorised with one
 * dissection subtree per rank (include/pgx_nd.h: pgx_nd_create_dist).  Residuals and observables are broadcast from rank 0
 * so that the replicas stay bitwise identical.  All calls are collective; results are identical on every rank.  Replaces
 * `mpirun -n N python obstacle_pg.py` with MUMPS distributing the factorisation (obstacle_pg.py:129-131). */
int pgx_create_lu_dist(const pgx_mesh* mesh, const pgx_problem* prob, pgx_comm* comm, int device, pgx_handle** out);
/* Owned VERTEX dofs of a local vector: each field block holds them at [offset, offset+count) (everything - all n vertex dofs - on
 * an unsharded handle).  P1: that is the whole field.  P2: the field block is [vertex dofs | edge dofs] and this call reports the
 * vertex part ONLY, on sharded and unsharded handles alike: a caller that slices the owned dofs of a P2 field must combine it
 * with pgx_owned_edge_range below (on an unsharded P2 handle: all vertices + all edges). */
int pgx_owned_range(const pgx_handle* h, int64_t* offset, int64_t* count);
/* P2 handles: the owned EDGE dofs of each field block as (offset within the field block, count); the vertex dofs are what
 * pgx_owned_range reports.  An edge belongs to the rank that owns its lower vertex; edge dofs are numbered by their lower vertex,
 * row by row, so the owned ones are contiguous.  Unsharded handles own every edge dof. */
int pgx_owned_edge_range(const pgx_handle* h, int64_t* offset, int64_t* count);
/* refresh the ghost entries of device `sol` and `sol_k` from their owners (after pgx_set_state / pgx_set_prev
 * with host arrays whose ghost entries are stale) */
int pgx_sync_ghosts(pgx_handle* h);

#ifdef __cplusplus
}
#endif
#endif
