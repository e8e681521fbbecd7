// pgx_sg.hip - example 02 (Signorini contact, 3-D linear elasticity, latent variable on the contact facets) behind the
// C ABI of include/pgx_sg.h.  Reference: examples/02_signorini/signorini_dolfinx.py (:146-153 sigma/eps, :199-249 spaces,
// measure and residual, :255-291 BCs and solver, :317-358 outer loop).
//
// x = [u_x | u_y | u_z | psi].  Everything that does not depend on the iterate is assembled ONCE on the device into the
// value array Jc of the mixed CSR pattern: the elasticity block A (constant-strain tetrahedra, 144 entries per cell) and
// the facet mass coupling +-M_G.  A residual is then one row-parallel pass over Jc (deterministic, no atomics) plus a
// facet kernel for the exp term; a Jacobian is "scale the A slots by alpha" plus 9 entries of D(psi) per contact facet.
#include <cstring>

#include "../../include/pgx_sg.h"
#include "pgx_mixed.h"
#include "pgx_scatter.h"

#define SG_MAXQ 16
struct SgQuad {
  double L[SG_MAXQ][3], w[SG_MAXQ];
  double N[SG_MAXQ][9];  // facet basis at the quadrature points: P1 = L; P2 = L_a (2 L_a - 1), then 4 L_a L_b for (0,1) (0,2) (1,2);
                         // quadrilateral facets: tensor Lagrange basis, lexicographic
  int nq;
  int g[3];              // local nodes spanning the (affine) facet: x = X[g0] + xi (X[g1] - X[g0]) + eta (X[g2] - X[g0])
};

static thread_local std::string g_sg_error;
static thread_local const pgx_sg_curved* g_sg_curved = nullptr;  // set by pgx_sg_create_curved around sg_create

struct pgx_sg_handle : MixedBase {
  int nv = 0, nc = 0, nf = 0, npsi = 0;  // nv = number of NODES (degree 2: vertices + edge midpoints)
  int npc = 4, npf = 3;                  // nodes per cell / per contact facet: 4 / 3 (degree 1), 10 / 6 (degree 2)
  SgQuad Q{};
  double alpha = 1.0, gap = 0.0, mu = 0.0, lmbda = 0.0;
  double *coords = nullptr, *gbc = nullptr, *bg = nullptr;
  double* fgeo = nullptr;  // order-2 geometry (pgx_sg_create_curved): [facet][point][2] = surface element, z of the curved facet; else nullptr
  int32_t *facets = nullptr, *fpsi = nullptr;
  // deterministic facet assembly (pgx_scatter.h): k_sg_exp parks [slot * nf + facet]; one thread per destination sums
  PgxScatter sc_D, sc_b;  // D(psi): 9 slots per facet -> CSR positions; <exp(psi), w>: 3 slots per facet -> residual rows
  double* stash = nullptr;  // [9 * nf]
  uint8_t *mask = nullptr, *kind = nullptr;
  double* Jc = nullptr;
  std::vector<int32_t> cverts;
  bool partitioned = false;  // distributed handle: this rank assembled the elasticity blocks of its slab of cells only
  int nc_owned = 0;
  void residual_dev(const double* xin, double* Fout) override;
  void jacobian_dev(const double* xin) override;
};

extern "C" const char* pgx_sg_last_error(const pgx_sg_handle* h) { return h ? h->err.c_str() : g_sg_error.c_str(); }

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
// elasticity block, once: A_e[(a,i),(b,j)] = vol (lambda G_ai G_bj + mu G_aj G_bi + mu delta_ij G_a.G_b)
__global__ __launch_bounds__(128) void k_sg_const_cells(int nc, const int32_t* __restrict__ cells, const double* __restrict__ coords,
                                                        double mu, double lmbda, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + 4 * (size_t)c;
  double X[4][3];
  for (int a = 0; a < 4; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[3 * (size_t)cv[a] + d];
  double J[3][3];  // columns = edge vectors
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 3; ++k) J[d][k] = X[k + 1][d] - X[0][d];
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  double inv[3][3];
  inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
  inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
  inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  // physical gradients G[a][d] = sum_k gref[a][k] inv[k][d], gref = [[-1,-1,-1],[1,0,0],[0,1,0],[0,0,1]]
  double G[4][3];
  for (int d = 0; d < 3; ++d) {
    G[1][d] = inv[0][d];
    G[2][d] = inv[1][d];
    G[3][d] = inv[2][d];
    G[0][d] = -(inv[0][d] + inv[1][d] + inv[2][d]);
  }
  const double vol = fabs(det) / 6.0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) {
      const double gg = G[a][0] * G[b][0] + G[a][1] * G[b][1] + G[a][2] * G[b][2];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          const double v = vol * (lmbda * G[a][i] * G[b][j] + mu * G[a][j] * G[b][i] + (i == j ? mu * gg : 0.0));
          stash[(size_t)((a * 3 + i) * 12 + (b * 3 + j)) * nc + c] = v;  // parked slot-major; summed per CSR position by pgx_scatter
        }
    }
}

// degree 2: A_e[(A,i),(B,j)] = |det J| sum_q w_q (lambda dN_A,i dN_B,j + mu dN_A,j dN_B,i + mu delta_ij dN_A . dN_B) with the
// 4-point degree-2 rule (exact: gradients of P2 functions are affine).  Local nodes: 0-3 vertices, 4-9 the edges
// (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); vertex functions L_a (2 L_a - 1), edge functions 4 L_a L_b.  900 entries per cell.
__global__ __launch_bounds__(64) void k_sg_const_cells_p2(int nc, const int32_t* __restrict__ cells, const double* __restrict__ coords,
                                                          double mu, double lmbda, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + 10 * (size_t)c;
  double X[4][3];
  for (int a = 0; a < 4; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[3 * (size_t)cv[a] + d];
  double J[3][3];
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 3; ++k) J[d][k] = X[k + 1][d] - X[0][d];
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  double inv[3][3];
  inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
  inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
  inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  double G1[4][3];  // gradients of the barycentric coordinates
  for (int d = 0; d < 3; ++d) {
    G1[1][d] = inv[0][d];
    G1[2][d] = inv[1][d];
    G1[3][d] = inv[2][d];
    G1[0][d] = -(inv[0][d] + inv[1][d] + inv[2][d]);
  }
  const int ea[6] = {0, 0, 0, 1, 1, 2}, eb[6] = {1, 2, 3, 2, 3, 3};
  const double qa = 0.5854101966249685, qb = 0.1381966011250105, wq = fabs(det) / 24.0;
  for (int A = 0; A < 10; ++A)
    for (int B = 0; B < 10; ++B) {
      double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int q = 0; q < 4; ++q) {
        double L[4];
        for (int a = 0; a < 4; ++a) L[a] = a == q ? qa : qb;
        double gA[3], gB[3];
        for (int d = 0; d < 3; ++d) {
          gA[d] = A < 4 ? (4.0 * L[A] - 1.0) * G1[A][d] : 4.0 * (L[ea[A - 4]] * G1[eb[A - 4]][d] + L[eb[A - 4]] * G1[ea[A - 4]][d]);
          gB[d] = B < 4 ? (4.0 * L[B] - 1.0) * G1[B][d] : 4.0 * (L[ea[B - 4]] * G1[eb[B - 4]][d] + L[eb[B - 4]] * G1[ea[B - 4]][d]);
        }
        const double gg = gA[0] * gB[0] + gA[1] * gB[1] + gA[2] * gB[2];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) acc[i][j] += wq * (lmbda * gA[i] * gB[j] + mu * gA[j] * gB[i] + (i == j ? mu * gg : 0.0));
      }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) stash[(size_t)((A * 3 + i) * 30 + (B * 3 + j)) * nc + c] = acc[i][j];
    }
}

// hexahedra (Q1 / Q2; any cell family with an affine map): A_e by quadrature from a TABLE of reference gradients,
// tab = [nq weights | nq x NPC x 3 reference gradients]; geometry from the four local nodes gn = (origin, +xi, +eta, +zeta):
// J = [X[g1]-X[g0] | X[g2]-X[g0] | X[g3]-X[g0]].  One thread per cell; gradients are recomputed per (A, B, q) so that nothing but the
// 3 x 3 accumulator lives in registers (Q2: 6561 entries per cell).
template <int NPC>
__global__ __launch_bounds__(64) void k_sg_const_cells_tab(int nc, const int32_t* __restrict__ cells, const double* __restrict__ coords,
                                                           double mu, double lmbda, int nq, const double* __restrict__ tab, int g0, int g1,
                                                           int g2, int g3, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int32_t* cv = cells + NPC * (size_t)c;
  const int gn[4] = {g0, g1, g2, g3};
  double X[4][3];
  for (int a = 0; a < 4; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[3 * (size_t)cv[gn[a]] + d];
  double J[3][3];
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < 3; ++k) J[d][k] = X[k + 1][d] - X[0][d];
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  double inv[3][3];
  inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
  inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
  inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  const double adet = fabs(det);
  const double* dN = tab + nq;
  constexpr int ND = 3 * NPC;
  for (int A = 0; A < NPC; ++A)
    for (int B = 0; B < NPC; ++B) {
      double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int q = 0; q < nq; ++q) {
        const double* ra = dN + ((size_t)q * NPC + A) * 3;
        const double* rb = dN + ((size_t)q * NPC + B) * 3;
        double gA[3], gB[3];
        for (int d = 0; d < 3; ++d) {
          gA[d] = ra[0] * inv[0][d] + ra[1] * inv[1][d] + ra[2] * inv[2][d];
          gB[d] = rb[0] * inv[0][d] + rb[1] * inv[1][d] + rb[2] * inv[2][d];
        }
        const double wq = tab[q] * adet;
        const double gg = gA[0] * gB[0] + gA[1] * gB[1] + gA[2] * gB[2];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) acc[i][j] += wq * (lmbda * gA[i] * gB[j] + mu * gA[j] * gB[i] + (i == j ? mu * gg : 0.0));
      }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) stash[(size_t)((A * 3 + i) * ND + (B * 3 + j)) * nc + c] = acc[i][j];
    }
}

// ISOPARAMETRIC cells (round 5: 10-node tetrahedra of an order-2 mesh, pgx_sg_create_curved): as k_sg_const_cells_tab, but |det J| and
// J^-1 come per (cell, quadrature point) from the caller's table geo[cell][q][10] - what a binding reads off the coordinate element -
// instead of once per cell from four nodes.  tab = [nq weights | nq x NPC x 3 reference gradients].
template <int NPC>
__global__ __launch_bounds__(64) void k_sg_const_cells_geo(int nc, double mu, double lmbda, int nq, const double* __restrict__ tab,
                                                           const double* __restrict__ geo, double* __restrict__ stash) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const double* dN = tab + nq;
  const double* gc = geo + (size_t)c * nq * 10;
  constexpr int ND = 3 * NPC;
  for (int A = 0; A < NPC; ++A)
    for (int B = 0; B < NPC; ++B) {
      double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int q = 0; q < nq; ++q) {
        const double* g = gc + (size_t)q * 10;  // g[0] = |det J|, g[1 + 3 k + d] = d xi_k / d x_d
        const double* ra = dN + ((size_t)q * NPC + A) * 3;
        const double* rb = dN + ((size_t)q * NPC + B) * 3;
        double gA[3], gB[3];
        for (int d = 0; d < 3; ++d) {
          gA[d] = ra[0] * g[1 + d] + ra[1] * g[4 + d] + ra[2] * g[7 + d];
          gB[d] = rb[0] * g[1 + d] + rb[1] * g[4 + d] + rb[2] * g[7 + d];
        }
        const double wq = tab[q] * g[0];
        const double gg = gA[0] * gB[0] + gA[1] * gB[1] + gA[2] * gB[2];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) acc[i][j] += wq * (lmbda * gA[i] * gB[j] + mu * gA[j] * gB[i] + (i == j ? mu * gg : 0.0));
      }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) stash[(size_t)((A * 3 + i) * ND + (B * 3 + j)) * nc + c] = acc[i][j];
    }
}

// facet mass coupling (+M on (u_z, psi), -M on (psi, u_z)) and b_g = <g, w>, once
template <int NPF>
__global__ __launch_bounds__(128) void k_sg_const_facets(int nf, const int32_t* __restrict__ facets, const int32_t* __restrict__ fpsi,
                                                         const double* __restrict__ coords, double gap, SgQuad Q,
                                                         double* __restrict__ stash /* [(2 NPF^2 + NPF) * nf]: matrix slots, then b_g */,
                                                         const double* __restrict__ fgeo /* curved facets: [f][q][2], else nullptr */) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int32_t* fv = facets + NPF * (size_t)f;
  double X[3][3];  // the three nodes spanning the affine facet
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[3 * (size_t)fv[Q.g[a]] + d];
  const double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
  const double e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
  const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
  const double area2 = sqrt(cx * cx + cy * cy + cz * cz);
  double Me[NPF][NPF], g[NPF];
  for (int a = 0; a < NPF; ++a) {
    g[a] = 0.0;
    for (int b = 0; b < NPF; ++b) Me[a][b] = 0.0;
  }
  for (int q = 0; q < Q.nq; ++q) {
    const double wd = Q.w[q] * (fgeo ? fgeo[2 * ((size_t)f * Q.nq + q)] : area2);
    const double zq = fgeo ? fgeo[2 * ((size_t)f * Q.nq + q) + 1] : Q.L[q][0] * X[0][2] + Q.L[q][1] * X[1][2] + Q.L[q][2] * X[2][2];
    for (int a = 0; a < NPF; ++a) {
      g[a] += wd * (zq - gap) * Q.N[q][a];
      for (int b = 0; b < NPF; ++b) Me[a][b] += wd * Q.N[q][a] * Q.N[q][b];
    }
  }
  for (int a = 0; a < NPF; ++a) {
    stash[(size_t)(2 * NPF * NPF + a) * nf + f] = g[a];
    for (int b = 0; b < NPF; ++b) {
      stash[(size_t)(a * NPF + b) * nf + f] = Me[a][b];               // row u_z(a), col psi(b)
      stash[(size_t)(NPF * NPF + a * NPF + b) * nf + f] = -Me[a][b];  // row psi(a), col u_z(b)
    }
  }
}

// kind: 0 = A slot (scaled by alpha), 1 = +-M_G slot, 2 = D slot (accumulated by k_sg_jac_D), 3 = BC diagonal, 4 = zeroed by BCs
__global__ void k_sg_jac_init(int64_t nnz, const uint8_t* __restrict__ kind, const double* __restrict__ Jc, double alpha,
                              double* __restrict__ Jv) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int t = kind[k];
  Jv[k] = t == 0 ? alpha * Jc[k] : t == 1 ? Jc[k] : t == 3 ? 1.0 : 0.0;
}

// parks (mode 0) D_e[a][b] = <exp(psi) N_a, N_b> for the Jacobian, (mode 1) b_exp[a] = <exp(psi), N_a> for the residual
template <int NPF>
__global__ __launch_bounds__(128) void k_sg_exp(int mode, int nf, int nu, const int32_t* __restrict__ facets,
                                                const int32_t* __restrict__ fpsi, const double* __restrict__ coords,
                                                const double* __restrict__ x, SgQuad Q, double* __restrict__ stash,
                                                const double* __restrict__ fgeo) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int32_t* fv = facets + NPF * (size_t)f;
  const int32_t* fp = fpsi + NPF * (size_t)f;
  double X[3][3];
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d) X[a][d] = coords[3 * (size_t)fv[Q.g[a]] + d];
  const double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
  const double e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
  const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
  const double area2 = sqrt(cx * cx + cy * cy + cz * cz);
  double pv[NPF], De[NPF][NPF], be[NPF];
  for (int a = 0; a < NPF; ++a) {
    pv[a] = x[nu + fp[a]];
    be[a] = 0.0;
    for (int b = 0; b < NPF; ++b) De[a][b] = 0.0;
  }
  for (int q = 0; q < Q.nq; ++q) {
    double pq = 0.0;
    for (int a = 0; a < NPF; ++a) pq += pv[a] * Q.N[q][a];
    const double e = Q.w[q] * (fgeo ? fgeo[2 * ((size_t)f * Q.nq + q)] : area2) * exp(pq);
    for (int a = 0; a < NPF; ++a) {
      be[a] += e * Q.N[q][a];
      for (int b = 0; b < NPF; ++b) De[a][b] += e * Q.N[q][a] * Q.N[q][b];
    }
  }
  if (mode == 0) {
    for (int a = 0; a < NPF; ++a)
      for (int b = 0; b < NPF; ++b) stash[(size_t)(a * NPF + b) * nf + f] = De[a][b];
  } else {
    for (int a = 0; a < NPF; ++a) stash[(size_t)a * nf + f] = be[a];
  }
}

// linear part of the residual from the constant values Jc (16 lanes per row), BC rows, - b_g
__global__ __launch_bounds__(256) void k_sg_resid_rows(int64_t ntot, int nu, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, const double* __restrict__ Jc,
                                                       const uint8_t* __restrict__ mask, const double* __restrict__ gbc,
                                                       const double* __restrict__ bg, const double* __restrict__ x,
                                                       const double* __restrict__ xk, double alpha, double* __restrict__ F) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  double a = 0.0;
  if (row < ntot) {
    const bool urow = row < nu;
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 16) {
      const int c = col[k];
      if (c < nu)
        a += (urow ? alpha : 1.0) * Jc[k] * (mask[c] ? gbc[c] : x[c]);
      else if (urow)
        a += Jc[k] * (x[c] - xk[c]);
    }
  }
  a += __shfl_xor(a, 8);
  a += __shfl_xor(a, 4);
  a += __shfl_xor(a, 2);
  a += __shfl_xor(a, 1);
  if (row < ntot && lane == 0) {
    if (row < nu)
      F[row] = mask[row] ? x[row] - gbc[row] : a;
    else
      F[row] = a - bg[row - nu];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------
extern "C" void pgx_sg_destroy(pgx_sg_handle* h) {
  if (!h) return;
  mx_release(h);
  delete h;
}

void pgx_sg_handle::residual_dev(const double* xin, double* Fout) {
  pgx_sg_handle* h = this;
  MxTimer t(h, 0);
  const int nu = 3 * h->nv;
  hipLaunchKernelGGL(k_sg_resid_rows, dim3((unsigned)((h->ntot * 16 + 255) / 256)), dim3(256), 0, h->st, h->ntot, nu, h->rowptr,
                     h->col, h->Jc, h->mask, h->gbc, h->bg, xin, h->xk, h->alpha, Fout);
  if (h->nf > 0) {
    if (h->npf == 6)
      hipLaunchKernelGGL(k_sg_exp<6>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 1, h->nf, nu, h->facets, h->fpsi, h->coords, xin,
                         h->Q, h->stash, h->fgeo);
    else if (h->npf == 4)
      hipLaunchKernelGGL(k_sg_exp<4>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 1, h->nf, nu, h->facets, h->fpsi, h->coords, xin,
                         h->Q, h->stash, h->fgeo);
    else if (h->npf == 9)
      hipLaunchKernelGGL(k_sg_exp<9>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 1, h->nf, nu, h->facets, h->fpsi, h->coords, xin,
                         h->Q, h->stash, h->fgeo);
    else
      hipLaunchKernelGGL(k_sg_exp<3>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 1, h->nf, nu, h->facets, h->fpsi, h->coords, xin,
                         h->Q, h->stash, h->fgeo);
    pgx_scatter_run(h->st, h->sc_b, h->stash, 1.0, 1, Fout);
  }
}
void pgx_sg_handle::jacobian_dev(const double* xin) {
  pgx_sg_handle* h = this;
  MxTimer t(h, 1);
  hipLaunchKernelGGL(k_sg_jac_init, dim3((unsigned)((h->nnz + 255) / 256)), dim3(256), 0, h->st, h->nnz, h->kind, h->Jc,
                     h->alpha, h->Jv);
  if (h->nf > 0) {
    if (h->npf == 6)
      hipLaunchKernelGGL(k_sg_exp<6>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 0, h->nf, 3 * h->nv, h->facets, h->fpsi,
                         h->coords, xin, h->Q, h->stash, h->fgeo);
    else if (h->npf == 4)
      hipLaunchKernelGGL(k_sg_exp<4>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 0, h->nf, 3 * h->nv, h->facets, h->fpsi,
                         h->coords, xin, h->Q, h->stash, h->fgeo);
    else if (h->npf == 9)
      hipLaunchKernelGGL(k_sg_exp<9>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 0, h->nf, 3 * h->nv, h->facets, h->fpsi,
                         h->coords, xin, h->Q, h->stash, h->fgeo);
    else
      hipLaunchKernelGGL(k_sg_exp<3>, dim3((h->nf + 127) / 128), dim3(128), 0, h->st, 0, h->nf, 3 * h->nv, h->facets, h->fpsi,
                         h->coords, xin, h->Q, h->stash, h->fgeo);
    pgx_scatter_run(h->st, h->sc_D, h->stash, 1.0, 1, h->Jv);
  }
  h->jac_valid = true;
}

// 1-D Lagrange basis on the equispaced nodes k / d (d = 1, 2): values l[0..d] and derivatives dl[0..d] at t
static void sg_lagrange1d(int d, double t, double* l, double* dl) {
  if (d == 1) {
    l[0] = 1.0 - t, l[1] = t;
    dl[0] = -1.0, dl[1] = 1.0;
  } else {
    l[0] = 2.0 * (t - 0.5) * (t - 1.0), l[1] = -4.0 * t * (t - 1.0), l[2] = 2.0 * t * (t - 0.5);
    dl[0] = 4.0 * t - 3.0, dl[1] = 4.0 - 8.0 * t, dl[2] = 4.0 * t - 1.0;
  }
}

// NPC / NPF: nodes per cell / per contact facet: tetrahedra 4 / 3 (degree 1), 10 / 6 (degree 2); hexahedra 8 / 4 (Q1), 27 / 9 (Q2).
// m->n_vertices = number of NODES.
template <int NPC, int NPF>
static int sg_create_impl(pgx_sg_handle* h, const pgx_sg_mesh* m, const pgx_sg_problem* p, pgx_comm* comm) {
  constexpr int ND = 3 * NPC, NE = ND * ND, NF2 = NPF * NPF, NFS = 2 * NF2 + NPF;  // cell dofs, cell slots, facet block, facet slots
  const int nv = m->n_vertices, nc = m->n_cells, nf = m->n_facets;
  h->npc = NPC, h->npf = NPF;
  if ((double)NE * nc * (sizeof(double) + sizeof(int32_t)) > 96e9) {  // stash + destination table of the one-pass constant-block assembly
    h->err = "mesh too large for the one-pass assembly of the elasticity block at this degree (" + std::to_string(NE) + " entries per cell)";
    return PGX_ENOMEM;
  }
  const int nu = 3 * nv;
  h->nv = nv, h->nc = nc, h->nf = nf;
  h->comm = comm;
  h->gap = p->gap;
  h->mu = p->E / (2.0 * (1.0 + p->nu));
  h->lmbda = p->E * p->nu / ((1.0 + p->nu) * (1.0 - 2.0 * p->nu));
  h->Q.nq = p->nq;
  for (int q = 0; q < p->nq; ++q) {
    const double X = p->qpts[2 * q], Y = p->qpts[2 * q + 1];
    h->Q.L[q][0] = 1.0 - X - Y, h->Q.L[q][1] = X, h->Q.L[q][2] = Y, h->Q.w[q] = p->qwts[q];
    const double* L = h->Q.L[q];
    if (NPF == 3) {
      for (int a = 0; a < 3; ++a) h->Q.N[q][a] = L[a];
    } else if (NPF == 6) {
      for (int a = 0; a < 3; ++a) h->Q.N[q][a] = L[a] * (2.0 * L[a] - 1.0);
      h->Q.N[q][3] = 4.0 * L[0] * L[1], h->Q.N[q][4] = 4.0 * L[0] * L[2], h->Q.N[q][5] = 4.0 * L[1] * L[2];
    } else {  // quadrilateral facets: tensor Lagrange basis of degree d on [0,1]^2, lexicographic (xi fastest)
      const int d = NPF == 4 ? 1 : 2;
      double lx[3], ly[3], dummy[3];
      sg_lagrange1d(d, X, lx, dummy);
      sg_lagrange1d(d, Y, ly, dummy);
      for (int iy = 0; iy <= d; ++iy)
        for (int ix = 0; ix <= d; ++ix) h->Q.N[q][iy * (d + 1) + ix] = lx[ix] * ly[iy];
    }
  }
  {
    const int d = (NPF == 4) ? 1 : (NPF == 9 ? 2 : 0);
    h->Q.g[0] = 0, h->Q.g[1] = d ? d : 1, h->Q.g[2] = d ? d * (d + 1) : 2;
  }
  for (size_t k = 0; k < NPC * (size_t)nc; ++k)
    if (m->cells[k] < 0 || m->cells[k] >= nv) {
      h->err = "cell vertex out of range";
      return PGX_EINVAL;
    }
  // psi dofs = contact vertices ordered by vertex id
  std::vector<int32_t> v2psi(nv, -1);
  for (size_t k = 0; k < NPF * (size_t)nf; ++k) {
    if (m->facets[k] < 0 || m->facets[k] >= nv) {
      h->err = "facet vertex out of range";
      return PGX_EINVAL;
    }
    v2psi[m->facets[k]] = 0;
  }
  h->cverts.clear();
  for (int v = 0; v < nv; ++v)
    if (v2psi[v] == 0) v2psi[v] = (int32_t)h->cverts.size(), h->cverts.push_back(v);
  const int npsi = (int)h->cverts.size();
  h->npsi = npsi;
  const int64_t ntot = (int64_t)nu + npsi;
  h->ntot = ntot;
  std::vector<uint8_t> hmask(nu, 0);
  std::vector<double> hg(nu, 0.0);
  for (int k = 0; k < p->n_bc; ++k) {
    const int d = p->bc_dofs[k];
    if (d < 0 || d >= nu) {
      h->err = "bc dof out of range";
      return PGX_EINVAL;
    }
    hmask[d] = 1;
    hg[d] = p->bc_vals ? p->bc_vals[k] : 0.0;
  }
  std::vector<int32_t> fpsi(NPF * (size_t)nf);
  for (size_t k = 0; k < fpsi.size(); ++k) fpsi[k] = v2psi[m->facets[k]];
  // entities: cells (12 mixed dofs: (a,i) -> i*nv + vertex a) and facets (6: u_z of 3 vertices, psi of 3 vertices)
  auto cell_dofs = [&](int c, int32_t md[ND]) {
    for (int a = 0; a < NPC; ++a)
      for (int i = 0; i < 3; ++i) md[a * 3 + i] = i * nv + m->cells[NPC * (size_t)c + a];
  };
  auto facet_dofs = [&](int f, int32_t md[2 * NPF]) {
    for (int a = 0; a < NPF; ++a) md[a] = 2 * nv + m->facets[NPF * (size_t)f + a], md[NPF + a] = nu + fpsi[NPF * (size_t)f + a];
  };
  std::vector<int64_t> dptr(ntot + 1, 0);
  for (int c = 0; c < nc; ++c) {
    int32_t md[ND];
    cell_dofs(c, md);
    for (int a = 0; a < ND; ++a) dptr[md[a] + 1]++;
  }
  for (int f = 0; f < nf; ++f) {
    int32_t md[2 * NPF];
    facet_dofs(f, md);
    for (int a = 0; a < 2 * NPF; ++a) dptr[md[a] + 1]++;
  }
  for (int64_t i = 0; i < ntot; ++i) dptr[i + 1] += dptr[i];
  std::vector<int64_t> dent(dptr[ntot]);  // entity id: cells [0,nc), facets [nc, nc+nf)
  {
    std::vector<int64_t> fill(dptr.begin(), dptr.end() - 1);
    for (int c = 0; c < nc; ++c) {
      int32_t md[ND];
      cell_dofs(c, md);
      for (int a = 0; a < ND; ++a) dent[fill[md[a]]++] = c;
    }
    for (int f = 0; f < nf; ++f) {
      int32_t md[2 * NPF];
      facet_dofs(f, md);
      for (int a = 0; a < 2 * NPF; ++a) dent[fill[md[a]]++] = (int64_t)nc + f;
    }
  }
  auto gather_row = [&](int64_t r, std::vector<int32_t>& tmp) {
    tmp.clear();
    for (int64_t q = dptr[r]; q < dptr[r + 1]; ++q) {
      const int64_t e = dent[q];
      if (e < nc) {
        int32_t md[ND];
        cell_dofs((int)e, md);
        tmp.insert(tmp.end(), md, md + ND);
      } else {
        int32_t md[2 * NPF];
        facet_dofs((int)(e - nc), md);
        tmp.insert(tmp.end(), md, md + 2 * NPF);
      }
    }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
  };
  std::vector<int32_t>& rowptr = h->h_rowptr;
  std::vector<int32_t>& col = h->h_col;
  rowptr.assign(ntot + 1, 0);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp;
    for (int64_t r = a; r < b; ++r) {
      gather_row(r, tmp);
      rowptr[r + 1] = (int32_t)tmp.size();
    }
  });
  int64_t tot = 0;
  for (int64_t r = 0; r < ntot; ++r) {
    tot += rowptr[r + 1];
    if (tot > 0x7fffffff) {
      h->err = "mixed matrix exceeds int32 nnz";
      return PGX_EINVAL;
    }
    rowptr[r + 1] = (int32_t)tot;
  }
  h->nnz = tot;
  col.resize(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    std::vector<int32_t> tmp;
    for (int64_t r = a; r < b; ++r) {
      gather_row(r, tmp);
      std::copy(tmp.begin(), tmp.end(), col.begin() + rowptr[r]);
    }
  });
  auto find = [&](int32_t r, int32_t c) -> int32_t {
    const int32_t* b = col.data() + rowptr[r];
    const int32_t* e = col.data() + rowptr[r + 1];
    return (int32_t)(std::lower_bound(b, e, c) - col.data());
  };
  std::vector<uint8_t> kind(tot);
  mx_par_for(ntot, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r)
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int32_t c = col[k];
        uint8_t t;
        if (r < nu && c < nu)
          t = (hmask[r] || hmask[c]) ? ((r == c && hmask[r]) ? 3 : 4) : 0;
        else if (r < nu)
          t = hmask[r] ? 4 : 1;
        else if (c < nu)
          t = hmask[c] ? 4 : 1;
        else
          t = 2;
        kind[k] = t;
      }
  });
  // Element partition of a distributed handle (round 4): the cells are cut into `size` slabs of equal count along the longest axis
  // of the mesh and every rank assembles the elasticity blocks of ITS slab only - what DOLFINx does with the owned cells of a
  // distributed mesh (signorini_dolfinx.py:283-291, `kind="mpi"`); ONE all-reduce sums the constant matrix at create time.  What a
  // Newton step assembles afterwards - a product with that matrix and the few thousand contact facets - stays replicated: it is
  // < 1 % of a step, and an all-reduce of the matrix values per step would cost more than it saves.
  std::vector<int32_t> own;
  {
    const char* e = pgx_tune("PGX_SG_PARTITION");
    const bool part = comm && comm->size > 1 && !(e && atoi(e) == 0);
    if (part) {
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int v = 0; v < nv; ++v)
        for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], m->coords[3 * (size_t)v + d]), hi[d] = std::max(hi[d], m->coords[3 * (size_t)v + d]);
      int ax = 0;
      for (int d = 1; d < 3; ++d)
        if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
      std::vector<double> key(nc);
      for (int c = 0; c < nc; ++c) {
        double z = 0.0;
        for (int a = 0; a < NPC; ++a) z += m->coords[3 * (size_t)m->cells[NPC * (size_t)c + a] + ax];
        key[c] = z;
      }
      std::vector<int32_t> order(nc);
      for (int c = 0; c < nc; ++c) order[c] = c;
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
      const int64_t c0 = (int64_t)nc * comm->rank / comm->size, c1 = (int64_t)nc * (comm->rank + 1) / comm->size;
      own.assign(order.begin() + c0, order.begin() + c1);
      std::sort(own.begin(), own.end());
      h->partitioned = true;
    } else {
      own.resize(nc);
      for (int c = 0; c < nc; ++c) own[c] = c;
    }
  }
  const int nco = (int)own.size();
  h->nc_owned = nco;
  // destination tables, slot-major like the stashes the kernels write: table[slot * n_entities + entity]
  std::vector<int32_t> d144((size_t)nco * NE), d18((size_t)nf * 2 * NF2), dD((size_t)nf * NF2), dbg((size_t)nf * NPF), dbe((size_t)nf * NPF);
  std::vector<int32_t> cown((size_t)nco * NPC);
  mx_par_for(nco, [&](int64_t a0, int64_t b0) {
    for (int64_t k = a0; k < b0; ++k) {
      const int c = own[k];
      int32_t md[ND];
      cell_dofs(c, md);
      for (int a = 0; a < NPC; ++a) cown[NPC * (size_t)k + a] = m->cells[NPC * (size_t)c + a];
      for (int a = 0; a < ND; ++a)
        for (int b = 0; b < ND; ++b) d144[(size_t)(a * ND + b) * nco + (size_t)k] = find(md[a], md[b]);
    }
  });
  for (int f = 0; f < nf; ++f) {
    int32_t md[2 * NPF];
    facet_dofs(f, md);
    for (int a = 0; a < NPF; ++a) {
      dbg[(size_t)a * nf + f] = fpsi[NPF * (size_t)f + a];
      dbe[(size_t)a * nf + f] = nu + fpsi[NPF * (size_t)f + a];
      for (int b = 0; b < NPF; ++b) {
        d18[(size_t)(a * NPF + b) * nf + f] = find(md[a], md[NPF + b]);
        d18[(size_t)(NF2 + a * NPF + b) * nf + f] = find(md[NPF + a], md[b]);
        dD[(size_t)(a * NPF + b) * nf + f] = find(md[NPF + a], md[NPF + b]);
      }
    }
  }
  std::vector<int32_t> nod(ntot);
  for (int v = 0; v < nv; ++v) nod[v] = nod[(size_t)nv + v] = nod[2 * (size_t)nv + v] = v;
  for (int k = 0; k < npsi; ++k) nod[(size_t)nu + k] = h->cverts[k];
  // device
  MXHIP(hipStreamCreate(&h->st));
  pgx_nd_matrix A{};
  A.n = ntot;
  A.rowptr = rowptr.data();
  A.col = col.data();
  A.n_nodes = nv;
  A.node_of_dof = nod.data();
  A.dim = 3;
  A.node_coords = m->coords;
  A.leaf_nodes = 0;
  if (const char* e = pgx_tune("PGX_ND_LEAF")) A.leaf_nodes = atoi(e);
  int rc = comm ? pgx_nd_create_dist(&A, comm, h->device, (void*)h->st, &h->lu) : pgx_nd_create(&A, h->device, (void*)h->st, &h->lu);
  if (rc) {
    h->err = std::string("direct solver: ") + pgx_nd_last_error(nullptr);
    h->lu = nullptr;
    return rc;
  }
  // [[alpha A, M_G^T], [-M_G, D(psi)]] with the rows of the latent block negated is the symmetric [[alpha A, M_G^T], [M_G, -D]]
  // (the Dirichlet rows AND columns of u are identity / zero): the LU takes it at half the flops (pgx_mixed.h lu_flip_from, pgx_nd.h);
  // PGX_SG_SYM=0 keeps the general LU of the matrix as UFL's derivative gives it (A/B)
  {
    const char* e = pgx_tune("PGX_SG_SYM");
    if (!(e && atoi(e) == 0)) {
      pgx_nd_set_symmetric(h->lu, 1);
      if (pgx_nd_is_symmetric(h->lu)) h->lu_flip_from = nu;
    }
  }
  int32_t* d_cells = nullptr;
  MXALLOC(h->coords, 3 * (size_t)nv);
  MXALLOC(h->facets, NPF * (size_t)nf);
  MXALLOC(h->fpsi, NPF * (size_t)nf);
  MXALLOC(h->stash, NF2 * (size_t)std::max(nf, 1));
  MXALLOC(h->mask, nu);
  MXALLOC(h->gbc, nu);
  MXALLOC(h->bg, npsi);
  MXALLOC(h->rowptr, ntot + 1);
  MXALLOC(h->col, tot);
  MXALLOC(h->kind, tot);
  MXALLOC(h->Jc, tot);
  MXALLOC(h->Jv, tot);
  if ((rc = mx_alloc_state(h))) return rc;
  MXHIP(hipMemcpy(h->coords, m->coords, sizeof(double) * 3 * nv, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->facets, m->facets, sizeof(int32_t) * NPF * (size_t)nf, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->fpsi, fpsi.data(), sizeof(int32_t) * fpsi.size(), hipMemcpyHostToDevice));
  if (g_sg_curved && nf > 0) {  // curved contact facets: surface element and z per quadrature point, read by every facet kernel
    MXALLOC(h->fgeo, 2 * (size_t)nf * p->nq);
    MXHIP(hipMemcpy(h->fgeo, g_sg_curved->facet_geo, sizeof(double) * 2 * (size_t)nf * p->nq, hipMemcpyHostToDevice));
  }
  MXHIP(hipMemcpy(h->mask, hmask.data(), nu, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->gbc, hg.data(), sizeof(double) * nu, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->rowptr, rowptr.data(), sizeof(int32_t) * (ntot + 1), hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->col, col.data(), sizeof(int32_t) * tot, hipMemcpyHostToDevice));
  MXHIP(hipMemcpy(h->kind, kind.data(), tot, hipMemcpyHostToDevice));
  MXHIP(hipMemsetAsync(h->Jc, 0, sizeof(double) * tot, h->st));
  MXHIP(hipMemsetAsync(h->bg, 0, sizeof(double) * npsi, h->st));
  {
    std::string e1 = pgx_scatter_build(dD.data(), (int64_t)NF2 * nf, tot, h->allocs, &h->sc_D);
    if (e1.empty()) e1 = pgx_scatter_build(dbe.data(), (int64_t)NPF * nf, ntot, h->allocs, &h->sc_b);
    if (!e1.empty()) {
      h->err = e1;
      return PGX_ENOMEM;
    }
  }
  // constant blocks, once, deterministic: park per entity, sum per destination; tables and stashes are temporary
  std::vector<void*> tmp;
  PgxScatter sc_c, sc_f, sc_g;
  std::string e1 = pgx_scatter_build(d144.data(), (int64_t)NE * nco, tot, tmp, &sc_c);
  if (e1.empty()) e1 = pgx_scatter_build(d18.data(), (int64_t)2 * NF2 * nf, tot, tmp, &sc_f);
  if (e1.empty()) e1 = pgx_scatter_build(dbg.data(), (int64_t)NPF * nf, npsi, tmp, &sc_g);
  double *st_c = nullptr, *st_f = nullptr;
  hipError_t e = e1.empty() ? hipMalloc((void**)&d_cells, sizeof(int32_t) * NPC * (size_t)std::max(nco, 1)) : hipErrorOutOfMemory;
  if (e == hipSuccess) e = hipMalloc((void**)&st_c, sizeof(double) * NE * (size_t)std::max(nco, 1));
  if (e == hipSuccess) e = hipMalloc((void**)&st_f, sizeof(double) * NFS * (size_t)std::max(nf, 1));
  if (e == hipSuccess) e = hipMemcpy(d_cells, cown.data(), sizeof(int32_t) * NPC * (size_t)nco, hipMemcpyHostToDevice);
  int rc_comm = PGX_OK;
  if (e == hipSuccess && nco > 0) {  // (a rank may own no cell at all - fewer cells than ranks: nothing to launch, zero contribution)
    // (the cell kernels below see the OWNED cells only: nco of them, compact)
    if (NPC == 4 && !g_sg_curved) {
      hipLaunchKernelGGL(k_sg_const_cells, dim3((nco + 127) / 128), dim3(128), 0, h->st, nco, d_cells, h->coords, h->mu, h->lmbda, st_c);
    } else if ((NPC == 10 || NPC == 4) && g_sg_curved) {
      // order-2 geometry: reference gradients of the field's shape functions at the caller's cell quadrature points - P2: ten (node
      // order of include/pgx_sg.h: 4 vertices, then the edges (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)), isoparametric; P1: the four
      // constant gradients of the barycentric coordinates on the quadratic cells - geometry per point from the caller's table
      const pgx_sg_curved* cv = g_sg_curved;
      const int nqc = cv->nq;
      std::vector<double> tab((size_t)nqc + (size_t)nqc * NPC * 3);
      static const double gref[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
      static const int ed[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
      for (int q = 0; q < nqc; ++q) {
        tab[q] = cv->qwts[q];
        const double X = cv->qpts[3 * q], Y = cv->qpts[3 * q + 1], Z = cv->qpts[3 * q + 2];
        const double L[4] = {1.0 - X - Y - Z, X, Y, Z};
        double* r = tab.data() + nqc + (size_t)q * NPC * 3;
        for (int a = 0; a < 4; ++a)
          for (int d = 0; d < 3; ++d) r[3 * a + d] = (NPC == 4 ? 1.0 : 4.0 * L[a] - 1.0) * gref[a][d];
        for (int k = 0; k < (NPC == 10 ? 6 : 0); ++k)
          for (int d = 0; d < 3; ++d) r[3 * (4 + k) + d] = 4.0 * (L[ed[k][0]] * gref[ed[k][1]][d] + L[ed[k][1]] * gref[ed[k][0]][d]);
      }
      double *d_tab = nullptr, *d_geo = nullptr;
      e = hipMalloc((void**)&d_tab, sizeof(double) * tab.size());
      if (e == hipSuccess) tmp.push_back(d_tab), e = hipMalloc((void**)&d_geo, sizeof(double) * 10 * (size_t)nqc * nc);
      if (e == hipSuccess) tmp.push_back(d_geo), e = hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(d_geo, cv->cell_geo, sizeof(double) * 10 * (size_t)nqc * nc, hipMemcpyHostToDevice);
      if (e == hipSuccess)  // (a curved handle is never partitioned: owned cell k IS cell k of the caller's table)
        hipLaunchKernelGGL(k_sg_const_cells_geo<(NPC == 10 ? 10 : 4)>, dim3((nco + 63) / 64), dim3(64), 0, h->st, nco, h->mu, h->lmbda, nqc,
                           d_tab, d_geo, st_c);
    } else if (NPC == 10) {
      hipLaunchKernelGGL(k_sg_const_cells_p2, dim3((nco + 63) / 64), dim3(64), 0, h->st, nco, d_cells, h->coords, h->mu, h->lmbda, st_c);
    } else {  // hexahedra: Gauss-Legendre (d + 1)^3 on [0,1]^3, tensor Lagrange reference gradients
      const int d = NPC == 8 ? 1 : 2, n1 = d + 1, nq3 = n1 * n1 * n1;
      const double gp2[2] = {0.5 - 0.5 / sqrt(3.0), 0.5 + 0.5 / sqrt(3.0)}, gw2[2] = {0.5, 0.5};
      const double gp3[3] = {0.5 - 0.5 * sqrt(0.6), 0.5, 0.5 + 0.5 * sqrt(0.6)}, gw3[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      const double* gp = d == 1 ? gp2 : gp3;
      const double* gw = d == 1 ? gw2 : gw3;
      std::vector<double> tab((size_t)nq3 + (size_t)nq3 * NPC * 3);
      int q = 0;
      for (int qz = 0; qz < n1; ++qz)
        for (int qy = 0; qy < n1; ++qy)
          for (int qx = 0; qx < n1; ++qx, ++q) {
            tab[q] = gw[qx] * gw[qy] * gw[qz];
            double lx[3], ly[3], lz[3], dx[3], dy[3], dz[3];
            sg_lagrange1d(d, gp[qx], lx, dx);
            sg_lagrange1d(d, gp[qy], ly, dy);
            sg_lagrange1d(d, gp[qz], lz, dz);
            for (int iz = 0; iz < n1; ++iz)
              for (int iy = 0; iy < n1; ++iy)
                for (int ix = 0; ix < n1; ++ix) {
                  double* r = tab.data() + nq3 + ((size_t)q * NPC + (size_t)(iz * n1 + iy) * n1 + ix) * 3;
                  r[0] = dx[ix] * ly[iy] * lz[iz], r[1] = lx[ix] * dy[iy] * lz[iz], r[2] = lx[ix] * ly[iy] * dz[iz];
                }
          }
      double* d_tab = nullptr;
      e = hipMalloc((void**)&d_tab, sizeof(double) * tab.size());
      if (e == hipSuccess) e = hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        tmp.push_back(d_tab);
        hipLaunchKernelGGL(k_sg_const_cells_tab<NPC>, dim3((nco + 63) / 64), dim3(64), 0, h->st, nco, d_cells, h->coords, h->mu, h->lmbda,
                           nq3, d_tab, 0, d, d * n1, d * n1 * n1, st_c);
      }
    }
    pgx_scatter_run(h->st, sc_c, st_c, 1.0, 0, h->Jc);
  }
  // sum of the slabs (fixed rank order: identical on every rank).  EVERY rank enters the collective, also one whose local steps
  // failed (its error is reported below, after the call): a rank that skipped it would leave the others waiting for the
  // transport's timeout instead of an error
  if (h->partitioned) rc_comm = comm->allreduce(h->st, h->Jc, (size_t)tot);
  if (e == hipSuccess) {
    if (nf > 0) {
      hipLaunchKernelGGL(k_sg_const_facets<NPF>, dim3((nf + 127) / 128), dim3(128), 0, h->st, nf, h->facets, h->fpsi, h->coords, h->gap,
                         h->Q, st_f, h->fgeo);
      pgx_scatter_run(h->st, sc_f, st_f, 1.0, 1, h->Jc);  // the +-M_G slots are disjoint from the elasticity slots
      pgx_scatter_run(h->st, sc_g, st_f + 2 * NF2 * (size_t)nf, 1.0, 0, h->bg);
    }
    e = hipStreamSynchronize(h->st);
  }
  hipFree(d_cells);
  hipFree(st_c);
  hipFree(st_f);
  for (void* q : tmp) hipFree(q);
  if (e != hipSuccess) {
    h->err = std::string("constant Jacobian blocks: ") + (e1.empty() ? hipGetErrorString(e) : e1.c_str());
    return PGX_EHIP;
  }
  if (rc_comm) {
    h->err = "sum of the partitioned elasticity blocks: " + comm->err;
    return rc_comm;
  }
  return PGX_OK;
}

static int sg_create(const pgx_sg_mesh* m, const pgx_sg_problem* p, pgx_comm* comm, int device, pgx_sg_handle** out) {
  if (!m || !p || !out || !m->coords || !m->cells || m->n_vertices <= 0 || m->n_cells <= 0 || m->n_facets < 0 ||
      (m->n_facets > 0 && !m->facets) || !p->qpts || !p->qwts || p->nq <= 0 || p->nq > SG_MAXQ ||
      (p->n_bc > 0 && !p->bc_dofs) || !(p->E > 0.0) || !(p->nu > -1.0 && p->nu < 0.5) || m->degree < 0 || m->degree > 2 || m->cell_type < 0 || m->cell_type > 1) {
    g_sg_error = "pgx_sg_create: bad arguments";
    return PGX_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_sg_error = "pgx_sg_create: no usable GPU (there is no CPU fallback)";
    return PGX_ENODEV;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_sg_error = "hipSetDevice failed";
    return PGX_EHIP;
  }
  pgx_sg_handle* h = new pgx_sg_handle();
  h->device = device;
  int rc;
  if (m->cell_type == 1)
    rc = (m->degree == 2) ? sg_create_impl<27, 9>(h, m, p, comm) : sg_create_impl<8, 4>(h, m, p, comm);
  else
    rc = (m->degree == 2) ? sg_create_impl<10, 6>(h, m, p, comm) : sg_create_impl<4, 3>(h, m, p, comm);
  if (rc) {
    g_sg_error = h->err;
    pgx_sg_destroy(h);
    return rc;
  }
  *out = h;
  return PGX_OK;
}

extern "C" int pgx_sg_create(const pgx_sg_mesh* m, const pgx_sg_problem* p, int device, pgx_sg_handle** out) {
  return sg_create(m, p, nullptr, device, out);
}
extern "C" int pgx_sg_create_curved(const pgx_sg_mesh* m, const pgx_sg_problem* p, const pgx_sg_curved* cv, int device,
                                    pgx_sg_handle** out) {
  if (!m || !cv || m->degree < 0 || m->degree > 2 || m->cell_type != 0 || cv->nq <= 0 || cv->nq > 512 || !cv->qpts || !cv->qwts || !cv->cell_geo ||
      (m->n_facets > 0 && !cv->facet_geo)) {
    g_sg_error = "pgx_sg_create_curved: order-2 geometry is implemented on tetrahedra (pgx_sg_mesh.cell_type = 0, degree 1 or 2) and "
                 "needs the cell rule and both geometry tables";
    return PGX_EINVAL;
  }
  g_sg_curved = cv;
  const int rc = sg_create(m, p, nullptr, device, out);
  g_sg_curved = nullptr;
  return rc;
}
extern "C" int pgx_sg_partition_info(const pgx_sg_handle* h, int64_t* owned_cells, int64_t* total_cells) {
  if (!h) return PGX_EINVAL;
  if (owned_cells) *owned_cells = h->nc_owned;
  if (total_cells) *total_cells = h->nc;
  return PGX_OK;
}
extern "C" int pgx_sg_lu_stats(const pgx_sg_handle* h, pgx_nd_stats* st) { return h ? pgx_nd_get_stats(h->lu, st) : PGX_EINVAL; }
extern "C" int pgx_sg_lu_is_symmetric(const pgx_sg_handle* h) { return h ? pgx_nd_is_symmetric(h->lu) : 0; }
extern "C" int pgx_sg_create_dist(const pgx_sg_mesh* m, const pgx_sg_problem* p, pgx_comm* comm, int device,
                                  pgx_sg_handle** out) {
  if (!comm) {
    g_sg_error = "pgx_sg_create_dist: null communicator";
    return PGX_EINVAL;
  }
  return sg_create(m, p, comm, device, out);
}

#define SGNEED(h)              \
  if (!(h)) return PGX_EINVAL; \
  if (hipSetDevice((h)->device) != hipSuccess) return PGX_EHIP

extern "C" int pgx_sg_num_dofs(const pgx_sg_handle* h, int64_t* ntot, int64_t* npsi) {
  if (!h) return PGX_EINVAL;
  if (ntot) *ntot = h->ntot;
  if (npsi) *npsi = h->npsi;
  return PGX_OK;
}
extern "C" int pgx_sg_contact_vertices(const pgx_sg_handle* h, int32_t* verts) {
  if (!h || !verts) return PGX_EINVAL;
  std::copy(h->cverts.begin(), h->cverts.end(), verts);
  return PGX_OK;
}
extern "C" int pgx_sg_set_state(pgx_sg_handle* h, const double* x) {
  SGNEED(h);
  return mx_in(h, h->x, x);
}
extern "C" int pgx_sg_get_state(pgx_sg_handle* h, double* x) {
  SGNEED(h);
  return mx_out(h, x, h->x);
}
extern "C" int pgx_sg_set_prev(pgx_sg_handle* h, const double* x) {
  SGNEED(h);
  return mx_in(h, h->xk, x);
}
extern "C" int pgx_sg_get_prev(pgx_sg_handle* h, double* x) {
  SGNEED(h);
  return mx_out(h, x, h->xk);
}
extern "C" int pgx_sg_advance_prev(pgx_sg_handle* h) {
  SGNEED(h);
  MXHIP(hipMemcpyAsync(h->xk, h->x, sizeof(double) * h->ntot, hipMemcpyDeviceToDevice, h->st));
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_sg_set_alpha(pgx_sg_handle* h, double a) {
  SGNEED(h);
  if (!(a > 0.0) || !std::isfinite(a)) {
    h->err = "alpha must be positive and finite";
    return PGX_EINVAL;
  }
  h->alpha = a;
  h->jac_valid = false;
  return PGX_OK;
}
extern "C" int pgx_sg_residual(pgx_sg_handle* h, const double* x, double* F, double* fnorm) {
  SGNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = mx_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  h->residual_dev(xd, h->F);
  if (fnorm) {
    int rc = mx_norm(h, h->F, fnorm);
    if (rc) return rc;
  }
  if (F) return mx_out(h, F, h->F);
  MXHIP(hipStreamSynchronize(h->st));
  return PGX_OK;
}
extern "C" int pgx_sg_jacobian_fill(pgx_sg_handle* h, const double* x) {
  SGNEED(h);
  const double* xd = h->x;
  if (x) {
    int rc = mx_in(h, h->xw, x);
    if (rc) return rc;
    xd = h->xw;
  }
  h->jacobian_dev(xd);
  MXHIP(hipStreamSynchronize(h->st));
  MXHIP(hipGetLastError());
  return PGX_OK;
}
extern "C" int pgx_sg_csr_export(pgx_sg_handle* h, int64_t* nrows, int64_t* nnz, int32_t* rowptr, int32_t* col,
                                 double* vals) {
  SGNEED(h);
  if (nrows) *nrows = h->ntot;
  if (nnz) *nnz = h->nnz;
  if (rowptr) std::copy(h->h_rowptr.begin(), h->h_rowptr.end(), rowptr);
  if (col) std::copy(h->h_col.begin(), h->h_col.end(), col);
  if (vals) {
    if (!h->jac_valid) {
      h->err = "pgx_sg_csr_export: no Jacobian has been filled";
      return PGX_ESTATE;
    }
    MXHIP(hipMemcpy(vals, h->Jv, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  }
  return PGX_OK;
}
extern "C" int pgx_sg_spmv(pgx_sg_handle* h, const double* x, double* y) {
  SGNEED(h);
  if (!x || !y) return PGX_EINVAL;
  if (!h->jac_valid) {
    h->err = "pgx_sg_spmv: no Jacobian has been filled";
    return PGX_ESTATE;
  }
  int rc = mx_in(h, h->r, x);
  if (rc) return rc;
  mx_spmv_dev(h, h->r, h->z);
  return mx_out(h, y, h->z);
}
extern "C" int pgx_sg_newton_solve(pgx_sg_handle* h, const pgx_snes_opts* opts, int* reason, int* its, int* lin_its) {
  SGNEED(h);
  if (!opts) return PGX_EINVAL;
  return opts->linesearch == 1 ? mx_newton_solve_bt(h, opts, reason, its, lin_its) : mx_newton_solve(h, opts, reason, its, lin_its);
}
extern "C" int pgx_sg_u_increment(pgx_sg_handle* h, double* out) {
  SGNEED(h);
  if (!out) return PGX_EINVAL;
  mx_axpby(h, 1.0, h->x, 0.0, h->r);
  mx_axpby(h, -1.0, h->xk, 1.0, h->r);
  return mx_norm(h, h->r, out, 3 * (int64_t)h->nv);
}
extern "C" int pgx_sg_profile(pgx_sg_handle* h, int enable, double ms[6]) {
  SGNEED(h);
  pgx_nd_timing(h->lu, enable, nullptr, nullptr);
  return mx_profile(h, enable, ms);
}
