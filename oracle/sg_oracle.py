"""CPU oracle for the LVPP Newton loop of example 02 (Signorini contact, 3-D linear elasticity, latent variable on the
contact surface).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (DOLFINx/PETSc/MUMPS absent, no reference tests or golden
data for this path) - restated from /root/reference/examples/02_signorini/signorini_dolfinx.py:

* mesh      : unit cube, nx x ny x nz cubes, 6 tetrahedra per cube around the diagonal v0-v7 (BASELINE.json config 5 asks
              for a tetrahedral P1 mesh; the script's native mesh is hexahedral, :376-383) [split pattern recalled from
              dolfinx create_box, not verifiable offline].  Contact surface z = 0, displacement surface z = 1 (:369-373).
* spaces    : u in (P1)^3 on the mesh, psi in P1 on the submesh of contact facets (:211-216), degree 1 here.
* residual  : (:236-249) with n_g = -e_z, g = x_z - gap, f = 0, facet quadrature degree 4 (:67-69,200-208):
                R_u   = alpha (sigma(u), eps(v)) - alpha (f, v) - <psi - psi_k, v.n_g>_Gamma
                R_psi = <u.n_g, w>_Gamma + <exp(psi), w>_Gamma - <g, w>_Gamma
              sigma = 2 mu eps + lambda tr(eps) I, mu = E/(2(1+nu)), lambda = E nu/((1+nu)(1-2nu)) (:146-153,234-235).
* Jacobian  : derivative: [[alpha A, +M_G (u_z, psi)], [-M_G (psi, u_z), D(psi)]], D = <exp(psi) N_a, N_b>.
* BCs       : u = (0, 0, disp) on the displacement surface (:255-268), DOLFINx lifting / set_bc contract.
* Newton    : SNES newtonls, linesearch none, atol = rtol = solver_tol (:331-332), PETSc defaults stol 1e-8, max_it 50
              (`newton_max_its` is accepted but never handed to the solver).
* outer     : it = 1..max_iterations; alpha = alpha_0 2^it (doubling) | alpha_0 + alpha_c it | constant (:323-328);
              solver_tol = 10 newton_tol for it < 2 (:330); stop when ||u - u_prev||_2 <= tol (vector 2-norm, :337-341);
              u_prev <- u, psi_k <- psi (:342-343).

DOF layout: x = [u_x (nv) | u_y (nv) | u_z (nv) | psi (n_contact vertices, ordered by vertex id)].
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import pg_oracle as O


def create_unit_cube_tets(nx, ny, nz):
    xs, ys, zs = np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), np.linspace(0, 1, nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (iz * (ny + 1) * (nx + 1) + iy * (nx + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    off = (nx + 1) * (ny + 1)
    v4, v5, v6, v7 = v0 + off, v1 + off, v2 + off, v3 + off
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.stack([np.stack(t, axis=1) for t in tets], axis=1).reshape(-1, 4).astype(np.int32)
    return coords, cells


def boundary_facets_where(coords, cells, pred):
    """triangles (vertex triples) of cell faces whose three vertices satisfy pred(coords) - all such faces are exterior
    for the planes z = 0 / z = 1 of the cube"""
    on = pred(coords)
    faces = np.concatenate([cells[:, [1, 2, 3]], cells[:, [0, 2, 3]], cells[:, [0, 1, 3]], cells[:, [0, 1, 2]]])
    sel = faces[on[faces].all(axis=1)]
    key = np.sort(sel, axis=1)
    _, idx = np.unique(key, axis=0, return_index=True)
    return sel[np.sort(idx)].astype(np.int32)


class SignoriniP1:
    def __init__(self, coords, cells, contact_facets, bc_vertices, E=2.0e4, nu=0.3, gap=0.0, disp=-0.25,
                 quadrature="tri_deg4_gj9", midside=None, cell_quadrature_degree=3):
        """midside (n_edges, 3), optional: ORDER-2 GEOMETRY (see SignoriniP2) - P1 fields on the quadratic cells and facets of a mesh of
        10-node tetrahedra; the cell integral by a rule of degree `cell_quadrature_degree` (UFL's estimate: 0 for the integrand of
        constant reference gradients + 3 for det J)."""
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.facets = np.ascontiguousarray(contact_facets, dtype=np.int32)
        self.nv, self.nc, self.nf = len(self.coords), len(self.cells), len(self.facets)
        self.cverts = np.unique(self.facets.ravel()).astype(np.int32)  # psi dof -> vertex
        self.npsi = len(self.cverts)
        self.v2psi = np.full(self.nv, -1, dtype=np.int64)
        self.v2psi[self.cverts] = np.arange(self.npsi)
        self.ntot = 3 * self.nv + self.npsi
        self.mu, self.lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
        self.gap, self.disp = float(gap), float(disp)
        bv = np.asarray(bc_vertices, dtype=np.int64)
        self.bc = np.concatenate([bv, self.nv + bv, 2 * self.nv + bv]).astype(np.int64)
        self.bc_vals = np.concatenate([np.zeros(len(bv)), np.zeros(len(bv)), np.full(len(bv), self.disp)])
        self.isbc = np.zeros(3 * self.nv, dtype=bool)
        self.isbc[self.bc] = True
        self.Xq, self.wq = O.load_quadrature(quadrature)
        self.Lq = np.stack([1 - self.Xq[:, 0] - self.Xq[:, 1], self.Xq[:, 0], self.Xq[:, 1]], axis=1)
        # elasticity: constant-strain tetrahedra
        x = self.coords[self.cells]
        J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2)  # columns
        det = np.linalg.det(J)
        invJ = np.linalg.inv(J)
        gref = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
        G = np.einsum("ak,ckd->cad", gref, invJ)  # physical gradients (nc,4,3)
        vol = np.abs(det) / 6.0
        mu, lm = self.mu, self.lmbda
        # A_e[(a,i),(b,j)] = vol * ( lambda G_ai G_bj + mu G_aj G_bi + mu delta_ij G_a.G_b )
        GG = np.einsum("cad,cbd->cab", G, G)
        Ae = (lm * np.einsum("cai,cbj->caibj", G, G) + mu * np.einsum("caj,cbi->caibj", G, G)
              + mu * np.einsum("cab,ij->caibj", GG, np.eye(3))) * vol[:, None, None, None, None]
        self.curved = midside is not None
        if self.curved:
            self.edges, self.cells10, self.facets6 = p2_numbering(self.cells, self.facets)
            self.geom_coords = np.concatenate([self.coords, np.ascontiguousarray(midside, dtype=np.float64)])
            self.cell_qpts, self.cell_qwts = tet_gauss_jacobi(cell_quadrature_degree)
            Lc = np.concatenate([1.0 - self.cell_qpts.sum(axis=1, keepdims=True), self.cell_qpts], axis=1)
            Jq = np.einsum("cad,qak->cqdk", self.geom_coords[self.cells10], _p2_tet_grad(Lc))
            detq = np.linalg.det(Jq)
            if np.any(detq.min(axis=1) * detq.max(axis=1) <= 0):
                raise ValueError("order-2 geometry: the cell map is not orientation preserving at every quadrature point")
            invq = np.linalg.inv(Jq)
            self.cell_geo = np.ascontiguousarray(np.concatenate([np.abs(detq)[..., None], invq.reshape(self.nc, -1, 9)], axis=2))
            Ae = np.zeros_like(Ae)
            for q in range(len(self.cell_qwts)):
                Gq = np.einsum("ak,ckd->cad", gref, invq[:, q])
                GGq = np.einsum("cad,cbd->cab", Gq, Gq)
                Ae += (self.cell_qwts[q] * np.abs(detq[:, q]))[:, None, None, None, None] * (
                    lm * np.einsum("cai,cbj->caibj", Gq, Gq) + mu * np.einsum("caj,cbi->caibj", Gq, Gq) + mu * np.einsum("cab,ij->caibj", GGq, np.eye(3)))
        rows = (self.cells[:, :, None, None, None] + self.nv * np.arange(3)[None, None, :, None, None])
        cols = (self.cells[:, None, None, :, None] + self.nv * np.arange(3)[None, None, None, None, :])
        rows, cols = np.broadcast_arrays(rows, cols)
        self.A = sp.coo_matrix((Ae.ravel(), (rows.ravel(), cols.ravel())), shape=(3 * self.nv, 3 * self.nv)).tocsr()
        # contact facets: area, mass matrix, g at quadrature points
        xf = self.coords[self.facets]
        self.farea2 = np.linalg.norm(np.cross(xf[:, 1] - xf[:, 0], xf[:, 2] - xf[:, 0]), axis=1)  # 2 * area
        self.wdet = self.farea2[:, None] * self.wq[None]  # (nf,nq)
        Mref = np.einsum("q,qa,qb->ab", self.wq, self.Lq, self.Lq)
        Me = self.farea2[:, None, None] * Mref[None]
        zq = np.einsum("qa,fa->fq", self.Lq, xf[:, :, 2])
        if self.curved:
            X6 = self.geom_coords[self.facets6]
            t = np.einsum("fad,qak->fqdk", X6, _p2_tri_grad(self.Lq))
            ds = np.linalg.norm(np.cross(t[..., 0], t[..., 1]), axis=2)
            self.wdet = ds * self.wq[None]
            Me = np.einsum("fq,qa,qb->fab", self.wdet, self.Lq, self.Lq)
            zq = np.einsum("qa,fa->fq", _p2_tri(self.Lq), X6[:, :, 2])
            self.facet_geo = np.ascontiguousarray(np.stack([ds, zq], axis=2))
        pf = self.v2psi[self.facets]
        self.pf = pf
        r = np.repeat(pf, 3, axis=1).ravel()
        c = np.tile(self.facets, (1, 3)).ravel()
        # MG[psi_a, vertex b] = <N_a, N_b>_Gamma  (npsi x nv)
        self.MG = sp.coo_matrix((Me.ravel(), (r, c)), shape=(self.npsi, self.nv)).tocsr()
        self.b_g = np.bincount(pf.ravel(), weights=((self.wdet * (zq - self.gap)) @ self.Lq).ravel(), minlength=self.npsi)
        self._rp = np.repeat(pf, 3, axis=1).ravel()
        self._cp = np.tile(pf, (1, 3)).ravel()

    def split(self, x):
        return x[: 3 * self.nv], x[3 * self.nv:]

    def exp_terms(self, psi, with_matrix=True):
        pq = psi[self.pf] @ self.Lq.T
        with np.errstate(over="ignore", under="ignore"):
            wE = self.wdet * np.exp(pq)
        b = np.bincount(self.pf.ravel(), weights=(wE @ self.Lq).ravel(), minlength=self.npsi)
        if not with_matrix:
            return b, None
        De = np.einsum("fq,qa,qb->fab", wE, self.Lq, self.Lq)
        return b, sp.coo_matrix((De.ravel(), (self._rp, self._cp)), shape=(self.npsi, self.npsi)).tocsr()

    def residual(self, x, xk, alpha):
        u, psi = self.split(x)
        psik = xk[3 * self.nv:]
        ut = u.copy()
        ut[self.bc] = self.bc_vals
        nv = self.nv
        Fu = alpha * (self.A @ ut)
        Fu[2 * nv:] += self.MG.T @ (psi - psik)  # -<psi - psi_k, v.n_g>, n_g = -e_z
        b_exp, _ = self.exp_terms(psi, with_matrix=False)
        Fp = -(self.MG @ ut[2 * nv:]) + b_exp - self.b_g
        Fu[self.bc] = u[self.bc] - self.bc_vals
        return np.concatenate([Fu, Fp])

    def jacobian(self, x, alpha):
        nv = self.nv
        _, D = self.exp_terms(x[3 * nv:])
        free = sp.diags((~self.isbc).astype(float))
        A = free @ (alpha * self.A) @ free + sp.diags(self.isbc.astype(float))
        Z = sp.csr_matrix((self.npsi, 2 * nv))
        B = sp.hstack([Z, self.MG], format="csr") @ free  # (npsi, 3nv): <N_a, N_b> on the u_z columns
        return sp.bmat([[A, (free @ B.T)], [-B, D]], format="csr")


def solve_contact_problem(prob: SignoriniP1, newton_tol=1e-6, max_iterations=25, alpha_scheme="doubling", alpha_0=1.0,
                          alpha_c=1.0, tol=1e-6, linear_solve=None, verbose=False, iterates=None):
    """Mirror of signorini_dolfinx.solve_contact_problem's loop (:317-358). Returns (x, it, iterations)."""
    x = np.zeros(prob.ntot)
    xk = x.copy()
    u_prev = np.zeros(3 * prob.nv)
    iterations = []
    it = 0
    for it in range(1, max_iterations + 1):
        alpha = alpha_0
        if alpha_scheme == "linear":
            alpha = alpha_0 + alpha_c * it
        elif alpha_scheme == "doubling":
            alpha = alpha_0 * 2**it
        solver_tol = 10 * newton_tol if it < 2 else newton_tol
        snes = O.SnesOptions(rtol=solver_tol, atol=solver_tol, stol=1e-8, max_it=50)
        xn, reason, its = O.newton_solve(prob, x, xk, alpha, snes, linear_solve)
        if reason <= 0:
            raise RuntimeError(f"SNES diverged at LVPP step {it}: reason {reason} after {its} its")
        x = xn
        iterations.append(its)
        nd = float(np.linalg.norm(x[: 3 * prob.nv] - u_prev))
        if iterates is not None:
            iterates.append(x.copy())
        if verbose:
            print(f"it={it} alpha={alpha:g} newton={its} reason={reason} |du|={nd:.3e} min psi {x[3 * prob.nv:].min():.2f}")
        if nd <= tol:
            break
        u_prev = x[: 3 * prob.nv].copy()
        xk = x.copy()
    return x, it, iterations


# ---------------------------------------------------------------------------------------------------------------------
# degree 2 (the reference's default, signorini_dolfinx.py:68-73): u in (P2)^3 on the tetrahedra, psi in P2 on the contact
# facets (:211-216 "degree of primal and latent space").  Conventions (this oracle's own; the HIP path follows them):
#   nodes     : the mesh vertices, then one node per edge, edges ordered lexicographically by their (min, max) vertex pair;
#               node coordinates of an edge node = the edge midpoint (affine cells)
#   tet       : local nodes 0-3 = vertices, 4-9 = edges (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
#   facet     : local nodes 0-2 = vertices, 3-5 = edges (0,1) (0,2) (1,2)
#   basis     : vertex a: L_a (2 L_a - 1);  edge (a, b): 4 L_a L_b
#   quadrature: cells - the 4-point degree-2 rule (exact: the integrand is a product of two affine gradients);
#               facets - the degree-4 rule of the P1 oracle (:67-69 quadrature_degree = 4)
# ---------------------------------------------------------------------------------------------------------------------
TET_EDGES = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)])
TRI_EDGES = np.array([(0, 1), (0, 2), (1, 2)])


def p2_numbering(cells, facets):
    """(n_vertices is implied) -> edges (ne,2), cells10 (nc,10), facets6 (nf,6) with edge node ids n_vertices + edge index."""
    nv = int(cells.max()) + 1
    pairs = np.sort(np.concatenate([cells[:, e] for e in TET_EDGES]), axis=1)
    edges, inv = np.unique(pairs, axis=0, return_inverse=True)
    nc = len(cells)
    c10 = np.concatenate([cells, nv + inv.reshape(6, nc).T], axis=1).astype(np.int32)
    key = edges[:, 0].astype(np.int64) * nv + edges[:, 1]
    fp = np.sort(np.concatenate([facets[:, e] for e in TRI_EDGES]), axis=1) if len(facets) else np.zeros((0, 2), dtype=np.int64)
    pos = np.searchsorted(key, fp[:, 0].astype(np.int64) * nv + fp[:, 1])
    f6 = np.concatenate([facets, nv + pos.reshape(3, len(facets)).T], axis=1).astype(np.int32) if len(facets) else np.zeros((0, 6), np.int32)
    return edges.astype(np.int32), c10, f6


def _p2_tri(L):
    """P2 basis on a triangle at barycentric points L (nq,3) -> (nq,6)"""
    return np.concatenate([L * (2 * L - 1), np.stack([4 * L[:, a] * L[:, b] for a, b in TRI_EDGES], axis=1)], axis=1)


def tet_gauss_jacobi(degree):
    """Collapsed-coordinate (Stroud conical product) Gauss-Jacobi rule on the reference tetrahedron, exact to `degree`:
    n = degree // 2 + 1 points per direction, (points (n^3, 3), weights summing to 1/6).  Basix's "GJ" scheme; its DEFAULT on
    simplices of low degree is Xiao-Gimbutas, whose tables are not available offline - any rule of the degree integrates the same
    polynomials exactly, the curved cells' rational integrands differ at the level of the quadrature error (parity unpinned)."""
    from scipy.special import roots_jacobi

    n = degree // 2 + 1
    x0, w0 = roots_jacobi(n, 0.0, 0.0)
    x1, w1 = roots_jacobi(n, 1.0, 0.0)
    x2, w2 = roots_jacobi(n, 2.0, 0.0)
    a, b, c = 0.5 * (x2 + 1.0), 0.5 * (x1 + 1.0), 0.5 * (x0 + 1.0)  # a: weight (1-a)^2, b: weight (1-b)
    wa, wb, wc = w2 / 8.0, w1 / 4.0, w0 / 2.0
    A, B, Cc = np.meshgrid(a, b, c, indexing="ij")
    W = wa[:, None, None] * wb[None, :, None] * wc[None, None, :]
    pts = np.stack([(Cc * (1 - B) * (1 - A)).ravel(), (B * (1 - A)).ravel(), A.ravel()], axis=1)
    return np.ascontiguousarray(pts), np.ascontiguousarray(W.ravel())


def _p2_tet_grad(L):
    """reference gradients of the ten P2 shape functions on the tetrahedron at barycentric points L (nq, 4) -> (nq, 10, 3);
    node order: 4 vertices, then the edges of TET_EDGES"""
    gref = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    G = np.empty((len(L), 10, 3))
    for a in range(4):
        G[:, a] = (4 * L[:, a] - 1)[:, None] * gref[a][None]
    for k, (a, b) in enumerate(TET_EDGES):
        G[:, 4 + k] = 4 * (L[:, a][:, None] * gref[b][None] + L[:, b][:, None] * gref[a][None])
    return G


def _p2_tri_grad(L):
    """reference gradients (d/dxi, d/deta) of the six P2 shape functions on the triangle at barycentric points L (nq, 3)"""
    gref = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    G = np.empty((len(L), 6, 2))
    for a in range(3):
        G[:, a] = (4 * L[:, a] - 1)[:, None] * gref[a][None]
    for k, (a, b) in enumerate(TRI_EDGES):
        G[:, 3 + k] = 4 * (L[:, a][:, None] * gref[b][None] + L[:, b][:, None] * gref[a][None])
    return G


class SignoriniP2:
    """Same interface as SignoriniP1 with nv := number of P2 nodes."""

    def __init__(self, coords, cells, contact_facets, bc_facets, E=2.0e4, nu=0.3, gap=0.0, disp=-0.25, quadrature="tri_deg4_gj9",
                 midside=None, cell_quadrature_degree=5):
        """midside (n_edges, 3), optional: the geometry's mid-edge nodes (edge order of p2_numbering: sorted (min, max) vertex pairs)
        of a mesh of 10-node tetrahedra - the reference's half sphere is one (lvpp/mesh_generation.py:88,158 `order=2`).  The
        discretisation is then ISOPARAMETRIC: cell map x(xi) = sum_a X_a N2_a(xi) over the ten nodes, Jacobians and physical
        gradients per quadrature point, cell integrals by a rule of degree `cell_quadrature_degree` (the reference leaves this
        integral's degree to UFL's estimator: 2 for the integrand on a simplex + 3 for det J of a quadratic tetrahedron = 5 as far
        as that can be told offline), facet integrals with the surface element |x_xi x x_eta| of the 6-node facets and g = x_z of
        the CURVED facet at the form's degree-4 points."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        facets = np.ascontiguousarray(contact_facets, dtype=np.int32)
        self.edges, self.cells10, self.facets6 = p2_numbering(cells, facets)
        self.nvert = len(coords)
        self.curved = midside is not None
        self.node_coords = np.concatenate([coords, 0.5 * (coords[self.edges[:, 0]] + coords[self.edges[:, 1]])
                                           if midside is None else np.ascontiguousarray(midside, dtype=np.float64)])
        self.coords, self.cells, self.facets = coords, cells, facets
        nn = self.nv = len(self.node_coords)
        self.nc, self.nf = len(cells), len(facets)
        self.cverts = np.unique(self.facets6.ravel()).astype(np.int32)  # psi dof -> node
        self.npsi = len(self.cverts)
        n2psi = np.full(nn, -1, dtype=np.int64)
        n2psi[self.cverts] = np.arange(self.npsi)
        self.ntot = 3 * nn + self.npsi
        self.mu, self.lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
        self.gap, self.disp = float(gap), float(disp)
        # Dirichlet nodes: every node of the displacement facets (vertices and edge nodes)
        _, _, bf6 = p2_numbering(cells, np.ascontiguousarray(bc_facets, dtype=np.int32))
        bn = np.unique(bf6.ravel()).astype(np.int64)
        self.bc_nodes = bn
        self.bc = np.concatenate([bn, nn + bn, 2 * nn + bn]).astype(np.int64)
        self.bc_vals = np.concatenate([np.zeros(len(bn)), np.zeros(len(bn)), np.full(len(bn), self.disp)])
        self.isbc = np.zeros(3 * nn, dtype=bool)
        self.isbc[self.bc] = True
        # elasticity block
        mu, lm = self.mu, self.lmbda
        Ae = np.zeros((self.nc, 10, 3, 10, 3))
        if not self.curved:
            a_, b_ = 0.5854101966249685, 0.1381966011250105
            Lq = np.full((4, 4), b_) + (a_ - b_) * np.eye(4)  # barycentric points of the degree-2 rule
            wq = np.full(4, 1.0 / 24.0)
            x = coords[cells]
            J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2)
            det = np.linalg.det(J)
            invJ = np.linalg.inv(J)
            gref = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
            G1 = np.einsum("ak,ckd->cad", gref, invJ)  # gradients of the barycentric coordinates (nc,4,3)
            for q in range(4):
                L = Lq[q]
                G = np.empty((self.nc, 10, 3))
                for a in range(4):
                    G[:, a] = (4 * L[a] - 1) * G1[:, a]
                for k, (a, b) in enumerate(TET_EDGES):
                    G[:, 4 + k] = 4 * (L[a] * G1[:, b] + L[b] * G1[:, a])
                GG = np.einsum("cad,cbd->cab", G, G)
                Ae += (wq[q] * np.abs(det))[:, None, None, None, None] * (
                    lm * np.einsum("cai,cbj->caibj", G, G) + mu * np.einsum("caj,cbi->caibj", G, G) + mu * np.einsum("cab,ij->caibj", GG, np.eye(3)))
        else:
            self.cell_qpts, self.cell_qwts = tet_gauss_jacobi(cell_quadrature_degree)
            Lc = np.concatenate([1.0 - self.cell_qpts.sum(axis=1, keepdims=True), self.cell_qpts], axis=1)
            dN = _p2_tet_grad(Lc)  # (nq,10,3)
            X10 = self.node_coords[self.cells10]  # (nc,10,3)
            J = np.einsum("cad,qak->cqdk", X10, dN)  # d x_d / d xi_k
            det = np.linalg.det(J)
            if np.any(det.min(axis=1) * det.max(axis=1) <= 0):  # (a cell may be negatively oriented as a whole: |det J| is what enters)
                raise ValueError("order-2 geometry: the cell map is not orientation preserving at every quadrature point")
            invJ = np.linalg.inv(J)  # [c,q,k,d] = d xi_k / d x_d
            self.cell_geo = np.ascontiguousarray(np.concatenate([np.abs(det)[..., None], invJ.reshape(self.nc, -1, 9)], axis=2))
            for q in range(len(self.cell_qwts)):
                G = np.einsum("ak,ckd->cad", dN[q], invJ[:, q])
                GG = np.einsum("cad,cbd->cab", G, G)
                Ae += (self.cell_qwts[q] * np.abs(det[:, q]))[:, None, None, None, None] * (
                    lm * np.einsum("cai,cbj->caibj", G, G) + mu * np.einsum("caj,cbi->caibj", G, G) + mu * np.einsum("cab,ij->caibj", GG, np.eye(3)))
        c10 = self.cells10
        rows = (c10[:, :, None, None, None] + nn * np.arange(3)[None, None, :, None, None])
        cols = (c10[:, None, None, :, None] + nn * np.arange(3)[None, None, None, None, :])
        rows, cols = np.broadcast_arrays(rows, cols)
        self.A = sp.coo_matrix((Ae.ravel(), (rows.ravel(), cols.ravel())), shape=(3 * nn, 3 * nn)).tocsr()
        # contact facets
        self.Xq, self.wq = O.load_quadrature(quadrature)
        L3 = np.stack([1 - self.Xq[:, 0] - self.Xq[:, 1], self.Xq[:, 0], self.Xq[:, 1]], axis=1)
        self.Nq = _p2_tri(L3)  # (nq,6)
        xf = coords[facets]
        f6 = self.facets6
        if not self.curved:
            farea2 = np.linalg.norm(np.cross(xf[:, 1] - xf[:, 0], xf[:, 2] - xf[:, 0]), axis=1)
            self.wdet = farea2[:, None] * self.wq[None]
            Mref = np.einsum("q,qa,qb->ab", self.wq, self.Nq, self.Nq)
            Me = farea2[:, None, None] * Mref[None]
            zq = np.einsum("qa,fa->fq", L3, xf[:, :, 2])
        else:
            X6 = self.node_coords[f6]  # (nf,6,3)
            dT = _p2_tri_grad(L3)  # (nq,6,2)
            t = np.einsum("fad,qak->fqdk", X6, dT)  # tangents x_xi, x_eta
            ds = np.linalg.norm(np.cross(t[..., 0], t[..., 1]), axis=2)  # surface element (nf,nq)
            self.wdet = ds * self.wq[None]
            Me = np.einsum("fq,qa,qb->fab", self.wdet, self.Nq, self.Nq)
            zq = np.einsum("qa,fa->fq", self.Nq, X6[:, :, 2])
            self.facet_geo = np.ascontiguousarray(np.stack([ds, zq], axis=2))
        pf = n2psi[f6]
        self.pf = pf
        r = np.repeat(pf, 6, axis=1).ravel()
        c = np.tile(f6, (1, 6)).ravel()
        self.MG = sp.coo_matrix((Me.ravel(), (r, c)), shape=(self.npsi, nn)).tocsr()
        self.b_g = np.bincount(pf.ravel(), weights=((self.wdet * (zq - self.gap)) @ self.Nq).ravel(), minlength=self.npsi)
        self._rp = np.repeat(pf, 6, axis=1).ravel()
        self._cp = np.tile(pf, (1, 6)).ravel()

    split = SignoriniP1.split
    residual = SignoriniP1.residual
    jacobian = SignoriniP1.jacobian

    def exp_terms(self, psi, with_matrix=True):
        pq = psi[self.pf] @ self.Nq.T
        with np.errstate(over="ignore", under="ignore"):
            wE = self.wdet * np.exp(pq)
        b = np.bincount(self.pf.ravel(), weights=(wE @ self.Nq).ravel(), minlength=self.npsi)
        if not with_matrix:
            return b, None
        De = np.einsum("fq,qa,qb->fab", wE, self.Nq, self.Nq)
        return b, sp.coo_matrix((De.ravel(), (self._rp, self._cp)), shape=(self.npsi, self.npsi)).tocsr()


# ---------------------------------------------------------------------------------------------------------------------
# hexahedra - the reference's NATIVE mesh (signorini_dolfinx.py:376-383: create_unit_cube(..., CellType.hexahedron), default
# 16 x 7 x 5) with Q_d elements, d = 1, 2 (d = 2 is the script's default degree).  Conventions (this oracle's own; the HIP path
# follows them): the mesh is the structured box grid, so the Q_d nodes are the lattice of (d nx + 1)(d ny + 1)(d nz + 1) points,
# numbered lexicographically (x fastest); a cell's / facet's local nodes are numbered lexicographically too; basis = tensor
# products of the 1-D Lagrange polynomials on the equispaced nodes k / d; cells by Gauss-Legendre (d + 1)^3 (exact on boxes),
# facets by Gauss-Legendre 3 x 3 (degree 5 >= the script's quadrature_degree = 4).
# ---------------------------------------------------------------------------------------------------------------------
def _lag1d(d, t):
    """values (len(t), d+1) and derivatives of the 1-D Lagrange basis on the nodes k/d"""
    xn = np.arange(d + 1) / d
    t = np.asarray(t, dtype=float)
    V = np.ones((len(t), d + 1))
    D = np.zeros((len(t), d + 1))
    for i in range(d + 1):
        for j in range(d + 1):
            if j != i:
                V[:, i] *= (t - xn[j]) / (xn[i] - xn[j])
        for m in range(d + 1):
            if m == i:
                continue
            term = np.full(len(t), 1.0 / (xn[i] - xn[m]))
            for j in range(d + 1):
                if j != i and j != m:
                    term *= (t - xn[j]) / (xn[i] - xn[j])
            D[:, i] += term
    return V, D


def _gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def hex_lattice(nx, ny, nz, d):
    """node coordinates (lexicographic, x fastest), cells (nc, (d+1)^3), bottom facets z = 0 and top facets z = 1 (nf, (d+1)^2)"""
    Nx, Ny, Nz = d * nx + 1, d * ny + 1, d * nz + 1
    Z, Y, X = np.meshgrid(np.linspace(0, 1, Nz), np.linspace(0, 1, Ny), np.linspace(0, 1, Nx), indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    nid = lambda gx, gy, gz: (gz * Ny + gy) * Nx + gx  # noqa: E731
    cz, cy, cx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cx, cy, cz = cx.ravel(), cy.ravel(), cz.ravel()
    loc = [(ix, iy, iz) for iz in range(d + 1) for iy in range(d + 1) for ix in range(d + 1)]
    cells = np.stack([nid(d * cx + ix, d * cy + iy, d * cz + iz) for ix, iy, iz in loc], axis=1).astype(np.int32)
    fy, fx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    fx, fy = fx.ravel(), fy.ravel()
    floc = [(ix, iy) for iy in range(d + 1) for ix in range(d + 1)]
    bottom = np.stack([nid(d * fx + ix, d * fy + iy, 0 * fx) for ix, iy in floc], axis=1).astype(np.int32)
    top = np.stack([nid(d * fx + ix, d * fy + iy, 0 * fx + Nz - 1) for ix, iy in floc], axis=1).astype(np.int32)
    return coords, cells, bottom, top


class SignoriniHex:
    """Same interface as SignoriniP1 with nv := number of Q_d nodes; cells must be parallelepipeds (the structured grid's are boxes)."""

    def __init__(self, nx, ny, nz, degree=2, E=2.0e4, nu=0.3, gap=0.0, disp=-0.25):
        d = self.degree = int(degree)
        coords, cells, bottom, top = hex_lattice(nx, ny, nz, d)
        self.node_coords, self.cells, self.facets, self.top = coords, cells, bottom, top
        nn = self.nv = len(coords)
        npc, npf = (d + 1) ** 3, (d + 1) ** 2
        self.nc, self.nf = len(cells), len(bottom)
        self.cverts = np.unique(bottom.ravel()).astype(np.int32)
        self.npsi = len(self.cverts)
        n2psi = np.full(nn, -1, dtype=np.int64)
        n2psi[self.cverts] = np.arange(self.npsi)
        self.ntot = 3 * nn + self.npsi
        self.mu, self.lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
        self.gap, self.disp = float(gap), float(disp)
        bn = np.unique(top.ravel()).astype(np.int64)
        self.bc_nodes = bn
        self.bc = np.concatenate([bn, nn + bn, 2 * nn + bn]).astype(np.int64)
        self.bc_vals = np.concatenate([np.zeros(len(bn)), np.zeros(len(bn)), np.full(len(bn), self.disp)])
        self.isbc = np.zeros(3 * nn, dtype=bool)
        self.isbc[self.bc] = True
        # elasticity block: reference gradients at the tensor Gauss points
        g, w = _gauss01(d + 1)
        V, D = _lag1d(d, g)
        loc = [(ix, iy, iz) for iz in range(d + 1) for iy in range(d + 1) for ix in range(d + 1)]
        x0 = coords[cells[:, 0]]
        J = np.stack([coords[cells[:, d]] - x0, coords[cells[:, d * (d + 1)]] - x0, coords[cells[:, d * (d + 1) ** 2]] - x0], axis=2)  # columns
        det = np.linalg.det(J)
        invJ = np.linalg.inv(J)
        Ae = np.zeros((self.nc, npc, 3, npc, 3))
        mu, lm = self.mu, self.lmbda
        for qz in range(d + 1):
            for qy in range(d + 1):
                for qx in range(d + 1):
                    gref = np.array([[D[qx, ix] * V[qy, iy] * V[qz, iz], V[qx, ix] * D[qy, iy] * V[qz, iz], V[qx, ix] * V[qy, iy] * D[qz, iz]]
                                     for ix, iy, iz in loc])  # (npc, 3)
                    G = np.einsum("ak,ckd->cad", gref, invJ)
                    GG = np.einsum("cad,cbd->cab", G, G)
                    wq = w[qx] * w[qy] * w[qz]
                    Ae += (wq * np.abs(det))[:, None, None, None, None] * (
                        lm * np.einsum("cai,cbj->caibj", G, G) + mu * np.einsum("caj,cbi->caibj", G, G) + mu * np.einsum("cab,ij->caibj", GG, np.eye(3)))
        rows = (cells[:, :, None, None, None] + nn * np.arange(3)[None, None, :, None, None])
        cols = (cells[:, None, None, :, None] + nn * np.arange(3)[None, None, None, None, :])
        rows, cols = np.broadcast_arrays(rows, cols)
        self.A = sp.coo_matrix((Ae.ravel(), (rows.ravel(), cols.ravel())), shape=(3 * nn, 3 * nn)).tocsr()
        # contact facets: 3 x 3 Gauss points on the reference square, bilinear / biquadratic facet basis
        gq, wq1 = _gauss01(3)
        Vf, _ = _lag1d(d, gq)
        self.fq = np.array([(gq[a], gq[b]) for b in range(3) for a in range(3)])  # (9, 2), xi fastest
        self.fw = np.array([wq1[a] * wq1[b] for b in range(3) for a in range(3)])
        self.Nq = np.array([[Vf[a, ix] * Vf[b, iy] for iy in range(d + 1) for ix in range(d + 1)] for b in range(3) for a in range(3)])  # (9, npf)
        xf0 = coords[bottom[:, 0]]
        e1, e2 = coords[bottom[:, d]] - xf0, coords[bottom[:, d * (d + 1)]] - xf0
        area = np.linalg.norm(np.cross(e1, e2), axis=1)
        self.wdet = area[:, None] * self.fw[None]
        Mref = np.einsum("q,qa,qb->ab", self.fw, self.Nq, self.Nq)
        Me = area[:, None, None] * Mref[None]
        pf = n2psi[bottom]
        self.pf = pf
        r = np.repeat(pf, npf, axis=1).ravel()
        c = np.tile(bottom, (1, npf)).ravel()
        self.MG = sp.coo_matrix((Me.ravel(), (r, c)), shape=(self.npsi, nn)).tocsr()
        zq = xf0[:, 2:3] + self.fq[None, :, 0] * e1[:, 2:3] + self.fq[None, :, 1] * e2[:, 2:3]
        self.b_g = np.bincount(pf.ravel(), weights=((self.wdet * (zq - self.gap)) @ self.Nq).ravel(), minlength=self.npsi)
        self._rp = np.repeat(pf, npf, axis=1).ravel()
        self._cp = np.tile(pf, (1, npf)).ravel()

    split = SignoriniP1.split
    residual = SignoriniP1.residual
    jacobian = SignoriniP1.jacobian
    exp_terms = SignoriniP2.exp_terms
