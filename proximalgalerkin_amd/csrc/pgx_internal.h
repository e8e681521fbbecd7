// Internal declarations shared by pgx_kernels.hip (device code + launch wrappers) and pgx_api.hip
// (host-side plan building, Newton / FGMRES / multigrid drivers, C ABI).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "pgx_scope.h"

// storage type of the multigrid D(psi) stencils.  They only feed the PRECONDITIONER (the exact operator's CSR D stays
// fp64), so fp32 would be admissible; measured on MI355X it made the smoother kernels 27 % SLOWER (same Krylov counts),
// so the stencils stay fp64.
#ifdef PGX_DSTEN_FLOAT
typedef float dsten_t;
#else
typedef double dsten_t;
#endif

#define PGX_MAX_NQ 16
#define PGX_BLOCK 256
#ifndef PGX_TILE_X
#define PGX_TILE_X 32  // tile of the fused multi-sweep smoothers (vertices). Measured ms per 2048^2 solve: 64x16 611,
                       // 32x16 564, 16x32 575, 64x8 582, 32x12 583, 32x24 586, 16x16 595, 32x8 597, 48x16 602, 128x8 693:
                       // the kernels are latency-bound, so LDS footprint (occupancy) beats halo redundancy
#define PGX_TILE_Y 16
#endif
#ifndef PGX_ROWMAP_TY
#define PGX_ROWMAP_TY 16  // tile rows of the row-mapped smoother k_st_smoothR (image 64 x (TY + 6) vertices)
#endif
#ifndef PGX_ROWMAP_BLOCK
#define PGX_ROWMAP_BLOCK 512  // threads per tile of k_st_smoothR: 8 waves, each owning every 8th image row
#endif

// Quadrature + P1 reference-element tables, passed BY VALUE as a kernel argument: they land in the
// kernarg segment and are read with scalar loads (wave-uniform), no constant-memory symbol to manage.
struct QuadTab {
  double N[PGX_MAX_NQ][3];  // P1 basis at quadrature points
  double w[PGX_MAX_NQ];     // weights (sum 1/2)
  double Mref[3][3];        // sum_q w_q N_a N_b
  double mref[3];           // sum_q w_q N_a
  int nq;
};

// 7-point stencil slots on a right-diagonal structured grid with row stride sx = nx+1:
//   0:(0,0) 1:(+1,0) 2:(-1,0) 3:(0,+1) 4:(0,-1) 5:(+1,+1) 6:(-1,-1)
// coefficient arrays are SoA: S[slot*n + v].  Links leaving the grid hold 0.
struct GridLevel {
  int nx, ny, n;             // cells per direction, vertices
  double *K, *M;             // [7*n] unmasked symmetric stencils, fixed at create
  // D(psi), refreshed every Newton step, symmetric-half storage [4*n]: slots 0:(0,0) 1:(+1,0) 2:(0,+1) 3:(+1,+1);
  // the three negative-direction links are the neighbours' positive ones (D is symmetric).
  dsten_t* Dh;
  float* Dh32;               // finest level only (else nullptr): single-precision copy of Dh for the interior tiles of the row-mapped
                             // smoother - a preconditioner inside FGMRES; the operator apply and every other kernel read Dh
  // On a uniform grid every interior vertex has the same K and M stencil: passed in the kernarg segment
  // instead of streaming 14 coefficient arrays. Verified on the host at create; 0 -> explicit arrays.
  int uniform;
  int interior_free;         // no Dirichlet dof strictly inside the grid (0 < i < nx, 0 < j < ny): interior tiles of the
                             // row-mapped smoother then need no mask tests.  Verified on the host at create.
  double Kc[7], Mc[7];
  uint8_t* mask;             // [n] 1 = Dirichlet dof of the u block
  double *xu, *xp, *xu2, *xp2;  // solution ping-pong
  double *bu, *bp;           // right-hand side
  double *ru, *rp;           // residual scratch
  // Single-precision V-cycle (pgx_mg32.hip; round 4).  On a level with f32 != 0 the cycle reads the D(psi) stencil as ONE float4 per
  // vertex - (D(0,0), D(+1,0), D(0,+1), D(+1,+1)), repacked from Dh with every Jacobian - and keeps its vectors as interleaved
  // (u, psi) float2: 40 B per vertex and smoother launch instead of 84.  The cycle is a preconditioner inside FGMRES (flexible);
  // the operator apply, the residuals and the Krylov space stay fp64.
  int f32;
  double4* Dd4;  // finest level only (else nullptr): the fp64 D stencil as ONE double4 per vertex for the matrix-free operator apply
                 // (two 16-byte loads per vertex instead of four 8-byte ones from four arrays); repacked with every Jacobian
  float4* Dq;
  float2 *xf, *xf2, *bf;
};

struct StConst {
  double K[7], M[7];
  int uniform;
};

// fused multigrid tail (k_mg_tail): level descriptors travel in the kernarg segment
#define PGX_TAIL_MAX 8
#define PGX_TAIL_VERTS 17000  // 129x129 vertices and below
struct TailLevel {
  int nx, ny, n;
  int interior_free;  // GridLevel::interior_free
  double ic[8];       // k_mg_tail2: alpha K (4) and M (4) interior stencils with the symmetric link pairs summed, set per launch
  const double *K, *M;
  const dsten_t* Dh;
  StConst sc;
  const uint8_t* mask;
  double *xu, *xp, *xu2, *xp2, *bu, *bp, *ru, *rp;
};
struct TailArgs {
  int nlev, nu, coarse_sweeps;
  double alpha, omega;
  TailLevel L[PGX_TAIL_MAX];
};
void pgxk_mg_tail(hipStream_t st, const TailArgs& A);
void pgxk_mg_tail_select(int v);  // 1 (default): k_mg_tail2; 0: the round-2 kernels

// ---- launch wrappers (pgx_kernels.hip). All asynchronous on `st`. -----------------------------
// stash: [4 * nc] scratch (element contributions parked at cell * 4 + a, summed per vertex through the v2c lists: no atomics)
void pgxk_bphi(hipStream_t st, int nc, int n, const int32_t* cells, const double* coords, const double* phi_q,
               QuadTab q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* bphi,
               const double* geo = nullptr /* order-2 geometry: [cell][point][5], the curved twins k_*_c of pgx_kernels.hip */);
void pgxk_gather_ent(hipStream_t st, int n, const int32_t* ptr, const int32_t* ent, const double* stash, double* out);
// mode 0: K, 1: M, 2: D(psi)
void pgxk_fill_rows(hipStream_t st, int mode, int n, size_t lds_bytes, const int32_t* rowptr, const int32_t* v2c_ptr,
                    const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells, const double* coords,
                    const double* psi, QuadTab q, double* out, const double* geo = nullptr);
// mode 0: y = J x ; 1: y = b - J x ; 2: y = x + omega*Binv*(b - J x) (collective Jacobi; first!=0: x taken as 0)
void pgxk_bspmv(hipStream_t st, int mode, int n, const int32_t* rowptr, const int32_t* colm, const double* K,
                const double* M, const double* D, double alpha, const double* xu, const double* xp, const double* bu,
                const double* bp, double omega, int first, double* yu, double* yp);
void pgxk_observables(hipStream_t st, int nc, int n, const int32_t* cells, const double* coords, const double* x,
                      const double* xk, double alpha, double f, QuadTab q, double* partials, int nblocks, double* out6,
                      int raw = 0, const double* geo = nullptr);
int pgxk_observables_blocks(int nc);

// vectors (length len)
void pgxk_axpy(hipStream_t st, size_t len, double a, const double* x, double* y);          // y += a x
void pgxk_scale_copy(hipStream_t st, size_t len, double a, const double* x, double* y);    // y = a x
void pgxk_set(hipStream_t st, size_t len, double a, double* y);
void pgxk_to_float(hipStream_t st, size_t len, const double* x, float* y);
// out[i] = V_i . w, i<nv (V_i = V + i*ldv).  partials: [PGX_RED_BLOCKS * nv] scratch
#define PGX_RED_BLOCKS 1024
// scale != nullptr: out[i] = scale->s[i] * (V_i . w)  (nv <= PGX_DOT_SCALE_MAX)
#define PGX_DOT_SCALE_MAX 64
struct PgxDotScale {
  double s[PGX_DOT_SCALE_MAX];
};
void pgxk_multidot(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* w, double* partials,
                   double* out, const PgxDotScale* scale = nullptr);
// w -= sum_i h[i] V_i   (h is a DEVICE pointer to nv doubles)
void pgxk_multiaxpy(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h, double* w);
// x = sum_i y[i] Z_i  (y device pointer); accumulate!=0 -> x += ...
void pgxk_lincomb(hipStream_t st, size_t len, int nv, const double* Z, size_t ldz, const double* y, double* x,
                  int accumulate);

// multigrid on stencil levels
void pgxk_csr_to_stencil(hipStream_t st, int n, int sx, const int32_t* rowptr, const int32_t* colm, const double* vals,
                         double* S);
void pgxk_csr_to_stencil_h(hipStream_t st, int n, int sx, const int32_t* rowptr, const int32_t* colm,
                           const double* vals, dsten_t* Sh, int frame_ny = 0);
void pgxk_rap7(hipStream_t st, const GridLevel& f, const double* Sf, const GridLevel& c, double* Sc);
void pgxk_rap7h(hipStream_t st, const GridLevel& f, const dsten_t* Sfh, const GridLevel& c, dsten_t* Sch);
void pgxk_st_apply(hipStream_t st, int mode, const GridLevel& L, double alpha, const double* xu, const double* xp,
                   const double* bu, const double* bp, double omega, int first, double* yu, double* yp);
void pgxk_restrict(hipStream_t st, const GridLevel& f, const double* ru, const double* rp, const GridLevel& c,
                   double* bu, double* bp);
void pgxk_prolong_add(hipStream_t st, const GridLevel& c, const double* cu, const double* cp, const GridLevel& f,
                      double* xu, double* xp);
void pgxk_coarse_mask(hipStream_t st, const GridLevel& c, uint8_t* mask_c, const GridLevel& f);
void pgxk_view_to_global(hipStream_t st, int ns, int n_view, int n_glob, int sx, int row0, int own0, int nown,
                         const double* in, double* out);
// fused V-cycle legs (nu = 2): two Jacobi sweeps per launch on LDS tiles, residual+restriction in one launch
void pgxk_st_smooth2(hipStream_t st, int post, const GridLevel& L, double alpha, const double* xu, const double* xp,
                     const GridLevel* C, const double* cu, const double* cp, const double* bu, const double* bp,
                     double omega, int remap, double* yu, double* yp);
void pgxk_st_spmv(hipStream_t st, const GridLevel& L, double alpha, const double* xu, const double* xp, int remap, double* yu,
                  double* yp, const float2* xf = nullptr /* the iterate as ONE interleaved (u, psi) float2 field instead of (xu, xp); uniform levels only */);
void pgxk_lincomb_f2(hipStream_t st, size_t n, int nv, const float2* Zf, size_t ldz, const double* y, double* xu, double* xp);
void pgxk_pack_d4(hipStream_t st, const GridLevel& L);  // Dd4 <- Dh
void pgxk_st_resid_restrict(hipStream_t st, const GridLevel& L, double alpha, const double* xu, const double* xp,
                            const double* bu, const double* bp, const GridLevel& C, int remap, double* cbu,
                            double* cbp);
// nnz-balanced CSR-stream kernel (blocks blk[b] .. blk[b+1] of <= PGX_BAL_CAP entries and <= 256 rows); bu != nullptr: b - J x
#ifndef PGX_BAL_CAP
#define PGX_BAL_CAP 1536  // entries per block of k_bspmv_bal; measured at 2048^2 P2, ms per apply on two boxes: 1024 1.46, 1280 1.37, 1536 1.31 / 1.13, 1792 - / 1.12, 2048 1.50, 3072 1.61, 4096 2.20
#endif
void pgxk_bspmv_bal(hipStream_t st, int n, int nblk, const int32_t* blk, const int32_t* rowptr, const int32_t* colm,
                    const double* K, const double* M, const uint8_t* code, const double* table, const double* D, double alpha,
                    const uint8_t* mask, const double* xu, const double* xp, const double* bu, const double* bp, int remap,
                    double* yu, double* yp, const float* Df = nullptr);
void pgxk_dict_assign(hipStream_t st, int64_t nnz, const double* K, const double* M, int ntab, const double* table, double tk,
                      double tm, uint8_t* code, int* fail, int cap, double* fail_v);
// CSR-stream form of y = Jx (256 rows per block through LDS); mask = Dirichlet flags of the u block
void pgxk_bspmv_stream(hipStream_t st, int n, size_t fill_lds_bytes, const int32_t* rowptr, const int32_t* colm,
                       const double* K, const double* M, const double* D, double alpha, const uint8_t* mask,
                       const double* xu, const double* xp, int remap, double* yu, double* yp);

// ---- P2 (pgx_p2.hip) -------------------------------------------------------------------------------
struct QuadTab2 {
  double L[PGX_MAX_NQ][3];      // P1 (barycentric) basis at quadrature points
  double N[PGX_MAX_NQ][6];      // P2 basis: 3 vertex functions l(2l-1), 3 edge functions 4 l_j l_k (edge i opposite vertex i)
  double dN[PGX_MAX_NQ][6][2];  // reference gradients
  double w[PGX_MAX_NQ];
  int nq;
};
// stash: [8 * nc] (bphi) / [16 * nc] (residual: u part then psi part) scratch, entries parked at cell * 8 + a and summed per
// dof through the dof -> (cell, local dof) lists of the P2 plan: no atomics, bitwise reproducible
void pgxk_bphi_p2(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords, const double* phi_q,
                  QuadTab2 q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* bphi,
                  const double* geo = nullptr /* [cell][point][5]: order-2 geometry, pgx_p2.hip */);
void pgxk_residual_p2_cells(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords,
                            const uint8_t* mask, const double* gbc, const double* x, const double* xk, double alpha,
                            double f, QuadTab2 q, const int32_t* v2c_ptr, const int32_t* v2c_ent, double* stash, double* F,
                            const double* geo = nullptr);
void pgxk_residual_final(hipStream_t st, int n, const uint8_t* mask, const double* gbc, const double* bphi,
                         const double* x, double* F);
void pgxk_fill_rows_p2(hipStream_t st, int mode, int n, size_t lds_bytes, const int32_t* rowptr,
                       const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cdofs,
                       const double* coords, const double* psi, QuadTab2 q, double* out, const double* geo = nullptr);
void pgxk_fill_rows_p1_Dp2(hipStream_t st, int nv, size_t lds_bytes, const int32_t* rowptr, const int32_t* v2c_ptr,
                           const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cdofs, const double* coords,
                           const double* psi, QuadTab2 q, double* out, const double* geo = nullptr);
void pgxk_p2_restrict(hipStream_t st, int nv, int n2, const int32_t* v2e_ptr, const int32_t* v2e, const uint8_t* mask1,
                      const double* ru2, const double* rp2, double* bu1, double* bp1);
void pgxk_p2_prolong_add(hipStream_t st, int nv, int n2, const int32_t* edge_ends, const double* cu, const double* cp,
                         double* xu, double* xp);
void pgxk_observables_p2_cells(hipStream_t st, int nc, int n, const int32_t* cdofs, const double* coords,
                               const double* x, const double* xk, double alpha, double f, QuadTab2 q, double* partials,
                               int nblocks, const double* geo = nullptr);
void pgxk_observables_final(hipStream_t st, int nblocks, const double* partials, double* out6);
void pgxk_observables_final_raw(hipStream_t st, int nblocks, const double* partials, double* out6);  // plain sums (sharded)
// vertex-star patch smoother of the P2 level (pgx_patch.hip)
void pgxk_patch_positions(hipStream_t st, int np, int NN, const int32_t* pdof, const int32_t* rowptr, const int32_t* colm,
                          int32_t* ppos);
void pgxk_patch_invert(hipStream_t st, int np, int NN, const int32_t* pdof, const int32_t* ppos, const double* K, const double* M,
                       const double* D, const uint8_t* mask, double alpha, void* pinv, int f32, int sym);
void pgxk_patch_sweep(hipStream_t st, int np, int NN, int nv, int nd, const int32_t* pdof, const int32_t* edge_ends,
                      const void* pinv, int f32, int sym, const double* ru, const double* rp, double omega, double* xu, double* xp,
                      double* su, double* sp);
size_t pgxk_patch_inverse_bytes(int np, int NN, int f32, int sym);  // sym: symmetric packing (float form only)
// structured P2 operator apply (pgx_p2st.hip): the 46 entries of an interior group's four rows (vertex: 19, three edges: 9 each)
struct P2StTab {
  int delta[46];             // neighbour dof = (isedge ? first edge dof of the group : vertex index of the group) + delta
  unsigned char isedge[46];
  int lofs[46];              // LDS form: offset of the neighbour in the block's vertex / edge tile relative to the thread's own slot
  int lds;                   // 1: every entry reaches at most one group row / column away (the tiles of k_p2st_apply_lds hold it)
  double aK[46], M[46];      // alpha K and M of the entry (uniform mesh: the same in every interior group)
  double kmax, mmax;
};
#define PGX_P2ST_BW 128      // groups per block of k_p2st_apply_lds
void pgxk_p2st_apply(hipStream_t st, const P2StTab& S, int nx, int nv, int i0, int ni, int j0, int nj, size_t G, const void* Dst, int f32,
                     const double* xu, const double* xp, const double* bu, const double* bp, double* yu, double* yp);
void pgxk_p2st_pack(hipStream_t st, int nx, int nv, int i0, int ni, int j0, int nj, size_t G, const int32_t* rowptr, const double* D,
                    void* Dst, int f32);
void pgxk_p2st_check(hipStream_t st, const P2StTab& S, int nx, int nv, int i0, int ni, int j0, int nj, double alpha_ref, double tol,
                     const int32_t* rowptr, const double* K, const double* M, int* fail);
void pgxk_p2_rows_csr(hipStream_t st, int nrows, const int32_t* rows, const int32_t* rowptr, const int32_t* colm, const double* K,
                      const double* M, const double* D, double alpha, const uint8_t* mask, const double* xu, const double* xp,
                      const double* bu, const double* bp, double* yu, double* yp);
// fused, atomic-free residual (+ optional D(psi) fill) for P1: see k_resid_fill_p1
void pgxk_resid_fill_p1(hipStream_t st, int write_d, int n, size_t lds_bytes, const int32_t* rowptr,
                        const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells,
                        const double* coords, const uint8_t* mask, const double* gbc, const double* bphi,
                        const double* x, const double* xk, double alpha, double f, QuadTab q, double* F, double* Dout,
                        const double* geo = nullptr);
void pgxk_resid_fill_grid(hipStream_t st, int write_d, const GridLevel& L, size_t lds_bytes, const int32_t* rowptr,
                          const int32_t* v2c_ptr, const int32_t* v2c_ent, const int32_t* v2c_pos, const int32_t* cells,
                          const double* coords, const uint8_t* mask, const double* gbc, const double* bphi, const double* x,
                          const double* xk, double alpha, double f, QuadTab q, double* F, double* Dout, int write_sh);
// write_sh: 0 = CSR rows only; 1 = CSR rows + the half-stencil L.Dh of the interior; 2 = half-stencil only (interior CSR rows NOT written)
// CGS2 with fused passes: (w' = w - V h1; [h2; |w'|^2] = [V,w']^T w') in one pass, then v = (w' - V h2)*scale
void pgxk_axpy_dot(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h1, double* w,
                   double* partials, double* out);
void pgxk_multiaxpy_scale(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h,
                          double scale, double* w);
// w -= V h and out[0] = |w'|^2 in one pass over the basis (selective CGS2: the lean second pass)
void pgxk_multiaxpy_norm(hipStream_t st, size_t len, int nv, const double* V, size_t ldv, const double* h, double* w,
                         double* partials, double* out);
// ---- single-precision V-cycle legs (pgx_mg32.hip) ----
// Dq <- Dh (values above 1e30 are clamped: an overshot Newton iterate must not put infinities into the preconditioner)
void pgxk_f_pack_d(hipStream_t st, const GridLevel& L);
// K (2 or 3) collective-Jacobi sweeps per launch on a single-precision level, out of place.
//   first != 0: S^K(0), xf unused; else S^K(xf + P x_c) with the coarse correction x_c = cf (float2: the next level is single
//   precision) or (cdu, cdp) (fp64 arrays), all three nullptr: none.  C = the coarse level (its nx), nullptr without correction.
//   b64u / b64p != nullptr: the right-hand side is read from these fp64 arrays and its float2 copy written to L.bf for the
//   launches that follow (finest level, first launch); else L.bf is read.
//   y64u / y64p != nullptr: the result is written as fp64 arrays (finest level, last launch); else to yf.
//   cbf or (cb64u, cb64p) != nullptr: the launch also restricts the residual of its result to the coarse level C, like
//   pgxk_f_resid_restrict (no coarse correction and a float2 result in that case).
void pgxk_f_smooth(hipStream_t st, int K, int first, const GridLevel& L, double alpha, const float2* xf, const double* b64u,
                   const double* b64p, const GridLevel* C, const float2* cf, const double* cdu, const double* cdp, double omega,
                   int remap, float2* yf, double* y64u, double* y64p, float2* cbf = nullptr, double* cb64u = nullptr,
                   double* cb64p = nullptr, double bscale = 1.0);  // bscale: factor applied to the fp64 right-hand side as it is read
// b_c = P^T (L.bf - J xf): to cbf (float2) or, when cb64u != nullptr, to the fp64 arrays (cb64u, cb64p) of an fp64 coarse level
void pgxk_f_resid_restrict(hipStream_t st, const GridLevel& L, double alpha, const float2* xf, const GridLevel& C, int remap,
                           float2* cbf, double* cb64u, double* cb64p);
// K (2 or 3) collective-Jacobi sweeps per launch; see k_st_smoothK
int pgxk_st_smooth6_ok(const GridLevel& L);
void pgxk_st_smoothK(hipStream_t st, int K, int post, const GridLevel& L, double alpha, const double* xu,
                     const double* xp, const GridLevel* C, const double* cu, const double* cp, const double* bu,
                     const double* bp, double omega, int remap, double* yu, double* yp);
